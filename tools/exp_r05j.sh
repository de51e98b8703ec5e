#!/bin/bash
# round 5, batch J: cache policy of the inverse kernel's 16-byte row loads (aux bits: 1 = sc0, 2 = nt, 16 = sc1), every batch size
O=gpurun_out/r05j
mkdir -p $O
for num in 8192 4096 2048 1024 512; do
  b2b=2; reps=150; [ $num -le 2048 ] && b2b=8 && reps=400
  for p in 1 2; do
    for v in base3 inl1 inl2 inl3 inl16 inl18 inl19; do
      echo "== r5_$v (process $p) $num polynomials"
      KB_PAIR=1 KB_B2B=$b2b ./tools/kbench_r5_$v $num $reps 20 40 | grep -E "^pair|^inverse"
    done
  done
done
