#!/bin/bash
# round 5 profile set (run on the GPU box from the repo root): rocprofv3 kernel trace + PMC passes at the bench batch (tools/profile.sh 1024)
# and the same at 8192 polynomials per launch (the HBM-streaming state).  Counters in their own passes, never with trace options.
set -u
export TMPDIR=/tmp
bash tools/profile.sh 1024 > /dev/null 2>&1
cp gpurun_out/prof/summary.txt gpurun_out/prof/summary_batch1024.txt
OUT=gpurun_out/prof8192
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_driver.py 8192 30 > $OUT/trace.log 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -- python3 tools/prof_driver.py 8192 2 > $OUT/pmc_$tag.log 2>&1
done
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A12 "k_forward15\|k_inverse15" $OUT/summary.txt | head -80
