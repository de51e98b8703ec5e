#!/bin/bash
# round 5, batch M: the inverse kernel's per-phase wave priorities (I1 = first round, I2 = the round that feeds the exchange, I3 = last round)
# retuned on the kernel with pre-landing (round 2 chose 3 / 0 / 2 on the kernel without it)
for num in 8192 1024; do
  b2b=2; reps=150; [ $num -le 2048 ] && b2b=8 && reps=400
  for p in 1 2; do
    for v in base6 pi301 pi303 pi312 pi202 pi300 pi201 pi313 pi102; do
      echo "== r5_$v (process $p) $num polynomials"
      KB_PAIR=1 KB_B2B=$b2b ./tools/kbench_r5_$v $num $reps 20 40 | grep -E "^pair|^inverse"
    done
  done
done
