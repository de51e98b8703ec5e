#!/bin/bash
# Round-3 batch A: the three structural experiments of VERDICT r02 item 1 on the n = 2^15 kernels (tools/kbench.hip,
# 1024 polynomials, 60-bit near-2^k class, warm; KB_PAIR adds forward -> inverse over the same buffer, KB_B2B back-to-back
# launches per sample as in bench.py).  r3base = round-2 kernels; c1 = n^-1 folded into the last inverse round's twiddles;
# a1 = c1 + forward loads issued in consumption order (0,16,1,17,...); b1 = c1 + polynomial tickets; ab = all three.
for p in 1 2 3; do
  for v in r3base c1 a1 b1 ab; do
    echo "== $v (process $p)"
    KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 1024 40 20 30
  done
done
for v in c1st b1st; do
  echo "== $v (stamped build: timeline)"
  KB_B2B=4 ./tools/kbench_$v 1024 20 20 100
done
