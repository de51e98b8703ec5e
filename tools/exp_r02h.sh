#!/bin/bash
# Round-2 batch H: inverse kernel walking the batch downwards (MI355NTT_INV_DESCENDING=1) vs upwards, on the
# forward -> inverse pair over one 256 MiB buffer (KB_PAIR) and on each kernel alone
for p in 1 2 3; do
  for v in asc desc; do
    echo "== $v (process $p)"
    KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 1024 40 20 30
  done
done
for v in asc desc; do
  echo "== $v num=512"; KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 512 40 20 30
  echo "== $v num=2048"; KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 2048 40 20 30
done
