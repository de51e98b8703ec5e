#!/bin/bash
# Round-2 batch E: barrier-free two-phase exchange in the forward kernel (k_forward15x) vs k_forward15; the library in
# the tree must have been built with -DMI355NTT_FWD_X=1 for the parity tests to exercise the new kernel.
cd "$(dirname "$0")/.."
OUT=gpurun_out/exp_r02e.txt
{
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -5
./tools/kbench_ship 1024 5 20 600 > /dev/null
for rep in 1 2 3 4; do
for v in ship fx; do
  echo "== $v (process $rep)"; KB_B2B=20 timeout 120 ./tools/kbench_$v 1024 15 20 200 | grep -E "forward"
done
done
} > $OUT 2>&1
tail -12 $OUT
