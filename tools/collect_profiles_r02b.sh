#!/bin/bash
# Round-2 evidence after the scheduling work (run on the GPU box from the repo root): kernel stats + overlap + one-step
# timeline of the bare train loop, kernel stats of bench.py.  Output under gpurun_out/ (copy into profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_loop.py 8 > gpurun_out/r02b_prof_train.log 2>&1 || exit 1
cp $(ls gpurun_out/prof_train/*/*kernel_stats.csv | head -1) gpurun_out/r02b_train_kernel_stats.csv
python3 tools/prof_summary.py gpurun_out/r02b_train_kernel_stats.csv 10 32 > gpurun_out/r02b_train_summary.txt
python3 tools/trace_overlap.py $(ls gpurun_out/prof_train/*/*kernel_trace.csv | head -1) 7 > gpurun_out/r02b_train_overlap.txt
python3 tools/step_timeline.py $(ls gpurun_out/prof_train/*/*kernel_trace.csv | head -1) > gpurun_out/r02b_step_timeline.txt
rm -rf gpurun_out/prof_train
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02b_prof_bench.log 2>&1 || exit 1
cp $(ls gpurun_out/prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/r02b_bench_kernel_stats.csv
python3 tools/prof_summary.py gpurun_out/r02b_bench_kernel_stats.csv 7 30 > gpurun_out/r02b_bench_summary.txt
rm -rf gpurun_out/prof_bench
tail -1 gpurun_out/r02b_prof_train.log; tail -1 gpurun_out/r02b_prof_bench.log | cut -c1-200
