#!/bin/bash
# round 5, batch L: cache policy of the forward kernel's 8-byte column loads (aux bits: 1 = sc0, 2 = nt, 16 = sc1), forward alone and pairs
for num in 8192 2048 1024; do
  b2b=2; reps=150; [ $num -le 2048 ] && b2b=8 && reps=400
  for p in 1 2; do
    for v in base5 fld1 fld2 fld16 fld17; do
      echo "== r5_$v (process $p) $num polynomials"
      KB_PAIR=1 KB_B2B=$b2b ./tools/kbench_r5_$v $num $reps 20 40 | grep -E "^pair|^forward"
    done
  done
done
