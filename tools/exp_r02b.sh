#!/bin/bash
# Round-2 batch B: steady-state VALU ceiling + process-to-process noise of the kernel timings (back-to-back launches)
cd "$(dirname "$0")/.."
OUT=gpurun_out/exp_r02b.txt
{
./tools/kbench_ship 1024 5 20 600 > /dev/null
./tools/ubench_ceiling 3000
for rep in 1 2 3 4 5 6; do
for v in ship fB fB4 fB12; do
  echo "== $v (process $rep)"; KB_B2B=20 ./tools/kbench_$v 1024 15 20 200 | grep -E "forward|inverse"
done
done
} > $OUT 2>&1
tail -3 $OUT
