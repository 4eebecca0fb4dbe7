#!/bin/bash
# Round-end evidence (run on the GPU box from the repo root): rocprofv3 kernel stats of bench.py and of the bare
# train loop, PMC passes (one counter group per pass) of the dominant 3x3 kernel and of the weight gradient.
# Output: gpurun_out/prof_*  (copy the summaries into profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -x
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
cp $(ls gpurun_out/prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/prof_bench_kernel_stats.csv
cp $(ls gpurun_out/prof_bench/*/*kernel_trace.csv | head -1) gpurun_out/prof_bench_kernel_trace.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_loop.py 8 > gpurun_out/prof_train.log 2>&1
cp $(ls gpurun_out/prof_train/*/*kernel_stats.csv | head -1) gpurun_out/prof_train_kernel_stats.csv
python3 tools/trace_overlap.py $(ls gpurun_out/prof_train/*/*kernel_trace.csv | head -1) 7 > gpurun_out/prof_train_overlap.txt
rm -rf gpurun_out/prof_bench gpurun_out/prof_train
for which in fwd16 wgrad16 fwd6 wgrad6; do
  kern=conv3x3; [ $which = wgrad6 -o $which = wgrad16 ] && kern=wgrad
  for spec in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "hit:TCC_HIT_sum TCC_MISS_sum" "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "lds:SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "g:GRBM_GUI_ACTIVE"; do
    name=${spec%%:*}; ctrs=${spec#*:}
    bash tools/pmc.sh pmc_${which}_$name $kern "$ctrs" -- python3 tools/one_kernel.py $which || exit 1
    rm -rf gpurun_out/pmc_${which}_$name
  done
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$which -- python3 tools/one_kernel.py $which > /dev/null 2>&1
  grep -E "$kern" $(ls gpurun_out/ks_$which/*/*kernel_stats.csv | head -1) | head -3
  rm -rf gpurun_out/ks_$which
done
