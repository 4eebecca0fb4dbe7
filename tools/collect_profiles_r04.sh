#!/bin/bash
# Round-4 evidence (GPU box, repo root): tools/collect_profiles_r04.sh <tag> [base] [batch]
# kernel stats + overlap + one-step timeline of the bare train loop (tools/train_loop.py), then the HBM traffic of a step
# from the TCC counters (two --pmc passes).  Output under gpurun_out/<tag>_* (copy the summaries into profiles/).
tag=${1:-r04}; base=${2:-hg2}; batch=${3:-32}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_loop.py 8 $batch $base > gpurun_out/${tag}_prof_train.log 2>&1 || exit 1
cp $(ls gpurun_out/prof_train/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_train_kernel_stats.csv
python3 tools/prof_summary.py gpurun_out/${tag}_train_kernel_stats.csv 10 40 > gpurun_out/${tag}_train_summary.txt
python3 tools/trace_overlap.py $(ls gpurun_out/prof_train/*/*kernel_trace.csv | head -1) 7 > gpurun_out/${tag}_train_overlap.txt
python3 tools/step_timeline.py $(ls gpurun_out/prof_train/*/*kernel_trace.csv | head -1) > gpurun_out/${tag}_step_timeline.txt
rm -rf gpurun_out/prof_train
tail -1 gpurun_out/${tag}_prof_train.log
steps=6
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/st_$c -- python3 tools/train_loop.py $steps $batch $base > gpurun_out/st_$c.log 2>&1 || exit 1
done
python3 tools/step_traffic.py $steps "$base + DSNT + JS, batch $batch" > gpurun_out/${tag}_step_traffic.txt
rm -rf gpurun_out/st_FETCH_SIZE gpurun_out/st_WRITE_SIZE
head -12 gpurun_out/${tag}_step_traffic.txt
