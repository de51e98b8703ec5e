#!/bin/bash
# round 5, batch N: start stagger of the transform kernels when a workgroup walks more than one polynomial (units of 2048 cycles between the
# 8 phase groups; 2 since round 2), retuned on the round-5 kernels
for num in 512 640 768 896 1024 2048 8192; do
  b2b=8; reps=400; [ $num -ge 4096 ] && b2b=2 && reps=150
  for p in 1 2; do
    for v in base7 sg0 sg1 sg3 sg4; do
      echo "== r5_$v (process $p) $num polynomials"
      KB_PAIR=1 KB_B2B=$b2b ./tools/kbench_r5_$v $num $reps 20 40 | grep -E "^pair|^inverse|^forward"
    done
  done
done
