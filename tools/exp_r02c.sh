#!/bin/bash
# Round-2 batch C: mad-chain butterflies vs the round-1 butterflies (same kernels otherwise), back-to-back launches
cd "$(dirname "$0")/.."
OUT=gpurun_out/exp_r02c.txt
{
./tools/kbench_mc 1024 5 20 600 > /dev/null
for rep in 1 2 3 4; do
for v in nomc mc mcB; do
  echo "== $v (process $rep)"; KB_B2B=20 ./tools/kbench_$v 1024 15 20 200 | grep -E "forward|inverse"
done
done
for v in nomc mc; do echo "== $v num=4096"; KB_B2B=10 ./tools/kbench_$v 4096 8 20 60 | grep -E "forward|inverse"; done
for v in nomc mc; do echo "== $v num=256"; KB_B2B=40 ./tools/kbench_$v 256 15 20 600 | grep -E "forward|inverse"; done
echo "== mcst"; ./tools/kbench_mcst 1024 20 20 400
} > $OUT 2>&1
tail -3 $OUT
