#!/bin/bash
# A/B kernel traces of the bare train loop under environment switches (run on the GPU box from the repo root):
#   tools/trace_ab.sh <tag> [VAR=value ...]      -> gpurun_out/ab_<tag>_overlap.txt, gpurun_out/ab_<tag>_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$tag -- python3 tools/train_loop.py 8 > gpurun_out/ab_$tag.log 2>&1 || exit 1
python3 tools/trace_overlap.py $(ls gpurun_out/ab_$tag/*/*kernel_trace.csv | head -1) 7 > gpurun_out/ab_${tag}_overlap.txt
cp $(ls gpurun_out/ab_$tag/*/*kernel_stats.csv | head -1) gpurun_out/ab_${tag}_stats.csv
python3 tools/step_timeline.py $(ls gpurun_out/ab_$tag/*/*kernel_trace.csv | head -1) > gpurun_out/ab_${tag}_timeline.txt
rm -rf gpurun_out/ab_$tag
tail -1 gpurun_out/ab_$tag.log; head -2 gpurun_out/ab_${tag}_overlap.txt
