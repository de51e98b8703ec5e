#!/bin/bash
# persistent grid of min(num, 256) against a balanced grid (every workgroup walks the same number of polynomials), n = 2^15
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
K=tools/kbench_r6grid
{
for num in 272 288 320 352 384 448 512 576 640 768 896; do
  rounds=$(( (num + 255) / 256 ))
  bal=$(( (num + rounds - 1) / rounds ))
  echo "== num $num: default grid $(( num < 256 ? num : 256 )), balanced $bal"
  KB_PAIR=1 KB_B2B=10 $K $num 15 0 60 | tail -1
  KB_GRID=$bal KB_PAIR=1 KB_B2B=10 $K $num 15 0 60 | tail -1
done
} 2>&1 | tee gpurun_out/grid_balance.txt
