#!/bin/bash
# round 4, batch B: where does the power go?  Sustained runs (about 3 s each) of the shipped n = 2^15 kernels and of ablation
# builds (results wrong, timing only), with rocm-smi package power / sclk sampled every 0.4 s in the background.
#   base: shipped; g: no global memory; t: no twiddle loads; x: no workgroup exchange; gt, gtx: combinations; c: no butterflies
#   (memory and LDS traffic only); gr: no global memory and no row staging
mkdir -p gpurun_out/r04a
for v in r4_base r4_ab_g r4_ab_t r4_ab_x r4_ab_gt r4_ab_gtx r4_ab_c r4_ab_gr; do
  echo "== $v"
  ( for i in $(seq 1 9); do sleep 0.4; rocm-smi --showpower --showclocks --csv 2>/dev/null | grep card0 | awk -F, '{print "      smi: sclk " $6 " power " $NF " W"}'; done ) &
  KB_PAIR=1 KB_B2B=8 ./tools/kbench_$v 1024 500 20 300 | grep -E "^pair|^forward|^inverse"
  wait
done
