#!/bin/bash
# Round-3 evidence (run on the GPU box from the repo root): tools/collect_profiles_r03.sh <tag> [bench]
# kernel stats + overlap + one-step timeline of the bare train loop; with "bench": kernel stats of bench.py too.
# Output under gpurun_out/<tag>_* (copy into profiles/).
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_loop.py 8 > gpurun_out/${tag}_prof_train.log 2>&1 || exit 1
cp $(ls gpurun_out/prof_train/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_train_kernel_stats.csv
python3 tools/prof_summary.py gpurun_out/${tag}_train_kernel_stats.csv 10 40 > gpurun_out/${tag}_train_summary.txt
python3 tools/trace_overlap.py $(ls gpurun_out/prof_train/*/*kernel_trace.csv | head -1) 7 > gpurun_out/${tag}_train_overlap.txt
python3 tools/step_timeline.py $(ls gpurun_out/prof_train/*/*kernel_trace.csv | head -1) > gpurun_out/${tag}_step_timeline.txt
rm -rf gpurun_out/prof_train
tail -1 gpurun_out/${tag}_prof_train.log
if [ "$2" = "bench" ]; then
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_prof_bench.log 2>&1 || exit 1
  cp $(ls gpurun_out/prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
  python3 tools/prof_summary.py gpurun_out/${tag}_bench_kernel_stats.csv 7 40 > gpurun_out/${tag}_bench_summary.txt
  rm -rf gpurun_out/prof_bench
  tail -1 gpurun_out/${tag}_prof_bench.log | cut -c1-300
fi
# PMC evidence of the round-3 kernels (tools/collect_profiles_r03.sh <tag> pmc): SQ issue accounting (tools/pmc_sweep.sh ->
# tools/pmc_account.py) and HBM traffic (FETCH_SIZE / WRITE_SIZE / L2 hit rate, one counter group per pass) of
#   conv3s (3x3 128->128 @64), wgrad3 (3x3 weight gradient), gemm1 (1x1 128->256 and 256->128 @64), wgrad1 (1x1 weight gradient)
if [ "$2" = "pmc" ]; then
  for spec in "c3s:conv3s:fwd16s:64 128 128 3" "wg3:wgrad3:wgrad16:64 128 128 3" "g1a:gemm1:fwd16:64 128 256 1" "g1b:gemm1:fwd16:64 256 128 1" "wg1:wgrad1:wgrad16:64 128 256 1"; do
    IFS=':' read name kern mode geo <<< "$spec"
    bash tools/pmc_sweep.sh ${tag}_$name $kern $mode $geo > /dev/null || exit 1
    python3 tools/pmc_account.py gpurun_out/pmcs_${tag}_$name.txt "$kern ($mode $geo, batch 32)" > gpurun_out/${tag}_pmc_$name.txt
    for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
      bash tools/pmc.sh ${tag}_t_$name $kern "$c" -- python3 tools/one_kernel.py $mode $geo | grep -v "^rc=" >> gpurun_out/${tag}_pmc_$name.txt
      rm -rf gpurun_out/${tag}_t_$name gpurun_out/${tag}_t_$name.log
    done
    timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 tools/one_kernel.py $mode $geo > /dev/null 2>&1
    grep -E "$kern" $(ls gpurun_out/ks/*/*kernel_stats.csv | head -1) | head -2 >> gpurun_out/${tag}_pmc_$name.txt
    rm -rf gpurun_out/ks
  done
  cat gpurun_out/${tag}_pmc_*.txt
fi
