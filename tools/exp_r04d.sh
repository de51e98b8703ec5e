#!/bin/bash
# round 4, batch D: canonicalisation by the sign of x - q formed with one 64-bit add (kbench_r4_cs) against the shipped compare + subtract pair
for p in 1 2 3 4; do for v in r4_base r4_cs; do echo "== $v (process $p)"; KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 1024 40 20 30 | grep -E "^pair|^forward|^inverse"; done; done
