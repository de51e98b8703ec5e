#!/bin/bash
# round 4, batch C: the HBM-streaming state (4096 polynomials = 1 GiB).  Shipped kernels against: both column halves' row loads of the inverse
# issued back to back (ml), start stagger 0 / 1 / 4 units instead of 2 (st0, st1, st4).  Two processes each.
for p in 1 2; do
  for v in r4_base r4_ml r4_st0 r4_st1 r4_st4; do
    echo "== $v process $p: 4096 polynomials"
    KB_PAIR=1 KB_B2B=2 ./tools/kbench_$v 4096 20 20 12 | grep -E "^pair|^forward|^inverse"
  done
done
