#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/exp_r02f.txt
{
./tools/kbench_ship 1024 5 20 600 > /dev/null
for rep in 1 2 3; do
for v in ship fx fxa fxb fxc; do
  echo "== $v (process $rep)"; KB_B2B=20 timeout 120 ./tools/kbench_$v 1024 15 20 200 | grep -E "forward"
done
done
echo "== fxst"; timeout 120 ./tools/kbench_fxst 1024 20 20 400 | grep -v inverse | head -42
echo "== fxst2"; timeout 120 ./tools/kbench_fxst2 1024 20 20 400 | grep -v inverse | head -12
} > $OUT 2>&1
