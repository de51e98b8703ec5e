#!/bin/bash
# Round-2 batch G: 16-byte column accesses (permlane32 swap) vs 8-byte ones; issue cost of the new instructions
cd "$(dirname "$0")/.."
OUT=gpurun_out/exp_r02g.txt
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -4
./tools/kbench_ship 1024 5 20 600 > /dev/null
for rep in 1 2 3 4; do
for v in nop16 ship; do
  echo "== $v (process $rep)"; KB_B2B=20 timeout 120 ./tools/kbench_$v 1024 15 20 200 | grep -E "forward|inverse"
done
done
./tools/ubench_issue 1000 | tail -6
} > $OUT 2>&1
