#!/bin/bash
# Evidence for the DSNT head's two train-step kernels at batch 1024 (GPU box, repo root): tools/head_profile.sh <tag>
#   -> gpurun_out/<tag>_head_kernels_b1024.txt: HIP-event timings (tools/bench_head.py), rocprofv3 kernel-trace averages, SQ issue
#      accounting and HBM traffic (FETCH_SIZE / WRITE_SIZE / L2 hits, one counter group per pass) per kernel
tag=${1:-r06}
out=gpurun_out/${tag}_head_kernels_b1024.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/bench_head.py 1024 2>&1 | grep -v amdgpu.ids > $out || exit 1
echo >> $out
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_head -- python3 tools/one_kernel.py head > /dev/null 2>&1 || exit 1
echo "rocprofv3 --kernel-trace --stats -- python3 tools/one_kernel.py head   (Name, Calls, TotalDurationNs, AverageNs, ...)" >> $out
grep -E "head_fwd_kernel|head_loss_grad_kernel" $(ls gpurun_out/ks_head/*/*kernel_stats.csv | head -1) >> $out
rm -rf gpurun_out/ks_head
for spec in "hf:head_fwd_kernel" "hg:head_loss_grad_kernel"; do
  IFS=':' read name kern <<< "$spec"
  bash tools/pmc_sweep.sh ${tag}_$name $kern head > /dev/null || exit 1
  echo >> $out
  python3 tools/pmc_account.py gpurun_out/pmcs_${tag}_$name.txt "$kern (16384 rows of 64 x 64 = batch 1024; 268 MB in + 268 MB out algorithmic)" >> $out
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    bash tools/pmc.sh ${tag}_t_$name $kern "$c" -- python3 tools/one_kernel.py head | grep -v "^rc=" >> $out
    rm -rf gpurun_out/${tag}_t_$name gpurun_out/${tag}_t_$name.log
  done
done
cat $out
