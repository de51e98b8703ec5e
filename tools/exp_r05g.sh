#!/bin/bash
# round 5, batch G: wave priority by age on the SIMD in the round that feeds the workgroup-wide exchange (the four waves of a SIMD reach
# the barrier up to 24 k cycles apart: section 2 of profiles/r05_streaming_overlap.txt).  ageF0 / ageI0 / ageFI: forward R1 / inverse R2' /
# both at priority = age (oldest 0 .. youngest 3); ageFm1: age - 1 (oldest two at 0).
O=gpurun_out/r05g
mkdir -p $O
for p in 1 2; do
  for v in base2 ageF0 ageI0 ageFI ageFm1; do
    echo "== r5_$v (process $p) 8192 polynomials"
    KB_PAIR=1 KB_B2B=2 ./tools/kbench_r5_$v 8192 150 20 40 | grep -E "^pair|^forward|^inverse"
  done
done
for p in 1 2; do
  for v in base2 ageFI ageFm1; do
    echo "== r5_$v (process $p) 1024 polynomials"
    KB_PAIR=1 KB_B2B=8 ./tools/kbench_r5_$v 1024 500 20 300 | grep -E "^pair|^forward|^inverse"
  done
done
echo "== r5_ageFIst (stamped) 8192 polynomials"
KB_B2B=2 ./tools/kbench_r5_ageFIst 8192 20 20 20 | grep -v "xcd \|phase \|wg "
