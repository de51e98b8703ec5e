#!/bin/bash
# round 5, batch I: touch-prefetch of the next polynomial into L2 and non-temporal accesses, HBM-streaming state.
#   tf1  forward: two vector touch loads at the loop top (a whole iteration ahead)      tf2 / tf3  forward: 128 / 64 scalar touch loads at the
#   start of the last round      tf4  forward: two vector touch loads behind the last round's last twiddle load
#   ti1  inverse: 64 scalar touch loads (second column half) behind the exchange      ti2  inverse: one vector touch load there
#   nts  forward row stores non-temporal      ntl  polynomial loads non-temporal (both kernels)
O=gpurun_out/r05i
mkdir -p $O
for p in 1 2; do
  for v in base3 tf1 tf2 tf3 tf4 ti1 ti2 tf4i2 nts ntl; do
    echo "== r5_$v (process $p) 8192 polynomials"
    KB_PAIR=1 KB_B2B=2 ./tools/kbench_r5_$v 8192 150 20 40 | grep -E "^pair|^forward|^inverse"
  done
done
for v in base3 tf4 ti2 tf4i2; do
  echo "== r5_$v 1024 polynomials"
  KB_PAIR=1 KB_B2B=8 ./tools/kbench_r5_$v 1024 500 20 300 | grep -E "^pair|^forward|^inverse"
done
