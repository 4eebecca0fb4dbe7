#!/bin/bash
# Broad SQ counter sweep of one kernel (one counter group per rocprofv3 pass, --pmc with --kernel-trace only):
#   tools/pmc_sweep.sh <tag> <kernel-substring> <one_kernel.py args...>     -> gpurun_out/pmcs_<tag>.txt
tag=$1; shift; kern=$1; shift
out=gpurun_out/pmcs_$tag.txt; : > $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
            "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" \
            "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  bash tools/pmc.sh pmcs_${tag}_$i $kern "$ctrs" -- python3 tools/one_kernel.py "$@" | grep -v "^rc=" >> $out || exit 1
  rm -rf gpurun_out/pmcs_${tag}_$i gpurun_out/pmcs_${tag}_$i.log
done
cat $out
