#!/bin/bash
# Round-3 batch B: round-2 kernels (r3base) against the round-3 state (c3 = c2 at the end of the round: n^-1 folded into the last inverse round, scaled
# twiddles read at their use site, forward loads in consumption order, ds_write2_b64 pair stores, lane id recomputed)
for p in 1 2 3; do
  for v in r3base c3; do
    echo "== $v (process $p)"
    KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 1024 40 20 30
  done
done
