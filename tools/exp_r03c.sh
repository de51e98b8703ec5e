#!/bin/bash
# Round-3 batch C: store cache policy of the frozen kernels once more (aux bits: 1 = sc0, 16 = sc1, 17 = write-through at system
# scope).  c3 = shipped default policy, v_wt = k_inverse15's stores written through, v_wtb = both kernels' stores, v_wt1 / v_wt16 =
# the single bits on the inverse.
for p in 1 2 3; do for v in c3 v_wt v_wtb v_wt1 v_wt16; do echo "== $v (process $p)"; KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 1024 40 20 30; done; done
