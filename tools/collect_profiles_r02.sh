#!/bin/bash
# Round-2 evidence (run on the GPU box from the repo root): rocprofv3 kernel stats of bench.py and of the bare train loop,
# SQ counter sweeps of the three hot kernel families (one counter group per pass), HBM traffic of the dominant kernel.
# Output under gpurun_out/ (copy the summaries into profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02_prof_bench.log 2>&1 || exit 1
cp $(ls gpurun_out/prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/r02_bench_kernel_stats.csv
python3 tools/prof_summary.py gpurun_out/r02_bench_kernel_stats.csv 7 30 > gpurun_out/r02_bench_summary.txt
rm -rf gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_loop.py 8 > gpurun_out/r02_prof_train.log 2>&1 || exit 1
cp $(ls gpurun_out/prof_train/*/*kernel_stats.csv | head -1) gpurun_out/r02_train_kernel_stats.csv
python3 tools/prof_summary.py gpurun_out/r02_train_kernel_stats.csv 10 30 > gpurun_out/r02_train_summary.txt
python3 tools/trace_overlap.py $(ls gpurun_out/prof_train/*/*kernel_trace.csv | head -1) 7 > gpurun_out/r02_train_overlap.txt
rm -rf gpurun_out/prof_train
for spec in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "hit:TCC_HIT_sum TCC_MISS_sum"; do
  name=${spec%%:*}; ctrs=${spec#*:}
  bash tools/pmc.sh r02_pmc_fwd16_$name conv3x3 "$ctrs" -- python3 tools/one_kernel.py fwd16 > gpurun_out/r02_pmc_fwd16_$name.txt || exit 1
  rm -rf gpurun_out/r02_pmc_fwd16_$name
done
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 tools/one_kernel.py fwd16 > /dev/null 2>&1
grep -E "conv3x3" $(ls gpurun_out/ks/*/*kernel_stats.csv | head -1) | head -2 > gpurun_out/r02_fwd16_kernel_stats.txt
rm -rf gpurun_out/ks
tail -1 gpurun_out/r02_prof_bench.log | cut -c1-300
cat gpurun_out/r02_pmc_fwd16_*.txt gpurun_out/r02_fwd16_kernel_stats.txt
