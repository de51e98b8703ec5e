#!/bin/bash
# Round-2 batch I: cache policy of the polynomial stores (aux bits: 1 = sc0, 2 = nt, 16 = sc1) against the end-of-kernel
# write-back that sits between back-to-back launches; pair = forward then inverse over the same 256 MiB
for p in 1 2; do
  for v in ship sta1 sta16 sta17 sta2 sta18; do
    echo "== $v (process $p)"
    KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 1024 40 20 30
  done
done
