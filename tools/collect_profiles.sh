#!/bin/bash
# Per-round evidence (GPU box, repo root): tools/collect_profiles.sh <tag> [base] [batch] [reg]      (base: hg1 | hg2 | hg8 | resnet34 ...)
# kernel stats + overlap + one-step timeline of the bare train loop (tools/train_loop.py), then the HBM traffic of a step
# from the TCC counters (two --pmc passes).  Output under gpurun_out/<tag>_* (copy the summaries into profiles/).
tag=${1:-r05}; base=${2:-hg2}; batch=${3:-32}; reg=${4:-js}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$2" != "pmc" ]; then
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_loop.py 8 $batch $base $reg > gpurun_out/${tag}_prof_train.log 2>&1 || exit 1
cp $(ls gpurun_out/prof_train/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_train_kernel_stats.csv
python3 tools/prof_summary.py gpurun_out/${tag}_train_kernel_stats.csv 10 40 > gpurun_out/${tag}_train_summary.txt
python3 tools/trace_overlap.py $(ls gpurun_out/prof_train/*/*kernel_trace.csv | head -1) 7 > gpurun_out/${tag}_train_overlap.txt
python3 tools/step_timeline.py $(ls gpurun_out/prof_train/*/*kernel_trace.csv | head -1) > gpurun_out/${tag}_step_timeline.txt
rm -rf gpurun_out/prof_train
tail -1 gpurun_out/${tag}_prof_train.log
steps=6
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/st_$c -- python3 tools/train_loop.py $steps $batch $base $reg > gpurun_out/st_$c.log 2>&1 || exit 1
done
python3 tools/step_traffic.py $steps "$base + DSNT (reg $reg), batch $batch" > gpurun_out/${tag}_step_traffic.txt
rm -rf gpurun_out/st_FETCH_SIZE gpurun_out/st_WRITE_SIZE
head -12 gpurun_out/${tag}_step_traffic.txt
fi

# PMC evidence of the 1x1 kernels and the dominant one (tools/collect_profiles.sh <tag> pmc ["c3s c3d c3f wg3" = a subset]): SQ issue accounting and HBM
# traffic (FETCH_SIZE / WRITE_SIZE / L2 hit rate, one counter group per pass) of
#   bwd1 (256 -> 128 with the folded BatchNorm backward, 128 -> 256 given), fwd1 (256 -> 128; 128 -> 256 + residual) and conv3s (3x3 128->128 @64)
if [ "$2" = "pmc" ]; then
  specs=("b1a:bwd1_kernel:bwd1a:64 256 128 1" "b1b:bwd1_kernel:bwd1:64 128 256 1" "f1a:fwd1_kernel:fwd1:64 256 128 1" "f1b:fwd1_kernel:fwd1:64 128 256 1" "s4f:stem4_fwd_kernel:stem4:0 0 0 0" "s4w:stem4_wgrad_kernel:stem4w:0 0 0 0" "c3s:conv3s:fwd16s:64 128 128 3" "c3d:conv3s:dgrad16s:64 128 128 3" "c3f:conv3s:fold3:64 128 128 3" "wg3:wgrad3_kernel:wgrad16:64 128 128 3")
  if [ -n "$3" ]; then specs=(); for n in $3; do for sp in "b1a:bwd1_kernel:bwd1a:64 256 128 1" "c3s:conv3s:fwd16s:64 128 128 3" "c3d:conv3s:dgrad16s:64 128 128 3" "c3f:conv3s:fold3:64 128 128 3" "wg3:wgrad3_kernel:wgrad16:64 128 128 3"; do [ "${sp%%:*}" = "$n" ] && specs+=("$sp"); done; done; fi
  for spec in "${specs[@]}"; do
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
