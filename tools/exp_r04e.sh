#!/bin/bash
# round 4, batch E: package power of a full-rate stream of ONE instruction (every SIMD, 4 waves each, 3 s per instruction), rocm-smi sampled
# every 0.4 s: which instructions of the butterfly are the expensive ones under the power cap?
# ops = indices into the table in main() of tools/ubench_issue.hip (the output names them): v_mul_lo_u32, v_mul_hi_u32, v_mad_u64_u32,
# v_lshl_add_u64, v_add_u32, subtract pair, v_lshlrev_b32, v_cndmask_b32 (a dependent chain), v_mov_b32, v_fma_f64
for op in 0 1 2 6 8 12 13 20 22 28; do
  ( for i in $(seq 1 7); do sleep 0.4; rocm-smi --showpower --showclocks --csv 2>/dev/null | grep card0 | awk -F, '{print "      smi: sclk " $6 " power " $NF " W"}'; done ) &
  ./tools/ubench_issue 20000 150 $op | grep -v "^#"
  wait
done
