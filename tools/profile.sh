#!/bin/bash
# rocprofv3 passes for the NTT kernels (run on the GPU box from the repo root).  Output: gpurun_out/prof/
# Counters are collected in their own passes, never together with the trace options gpurun refuses.
set -u
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
BATCH=${1:-1024}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_driver.py $BATCH 200 > $OUT/trace.log 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -- python3 tools/prof_driver.py $BATCH 4 > $OUT/pmc_$tag.log 2>&1
done
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
