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
