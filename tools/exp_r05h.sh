#!/bin/bash
# round 5, batch H: the SMU's throttler residencies of the real kernels WHILE THEY RUN at 8192 polynomials (batch B sampled whole
# processes, host-side set-up included): kbench in the background, smi_watch from second 5 to second 9 of a run whose timed loops
# start at about second 3.  Shipped kernels and four ablations.
O=gpurun_out/r05h
mkdir -p $O
for v in base2 g c tx s; do
  [ -x tools/kbench_r5_$v ] || continue
  echo "== r5_$v 8192 polynomials, smi_watch over seconds 5..9 of the run"
  ( KB_PAIR=1 KB_B2B=2 ./tools/kbench_r5_$v 8192 900 20 40 | grep -E "^pair|^forward|^inverse" ) &
  K=$!
  sleep 5
  ./tools/smi_watch 100 40 > $O/smi_$v.log 2>&1
  python3 tools/pviol.py $O/smi_$v.log 0
  wait $K
done
