#!/bin/bash
# Round-2 batch D: spill-free kernels (scalar modulus index, thread-derived values rebuilt at use) vs the previous build
cd "$(dirname "$0")/.."
OUT=gpurun_out/exp_r02d.txt
{
./tools/kbench_ship 1024 5 20 600 > /dev/null
for rep in 1 2 3 4; do
for v in mcB ship; do
  echo "== $v (process $rep)"; KB_B2B=20 ./tools/kbench_$v 1024 15 20 200 | grep -E "forward|inverse"
done
done
echo "== st1"; ./tools/kbench_st1 1024 20 20 400
echo "== st2"; ./tools/kbench_st2 1024 20 20 400 | grep -v inverse | head -40
} > $OUT 2>&1
tail -3 $OUT
