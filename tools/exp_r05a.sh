#!/bin/bash
# round 5, batch A: the n = 2^15 kernels in the HBM-streaming state (8192 polynomials = 2 GiB).
#   ablation builds (results wrong, timing only): g no global memory, t no twiddle loads, x no workgroup exchange, tx, gtx,
#   c no butterflies (memory + LDS only), s no result stores, kN: the first N of the 32 row/column loads of every polynomial
#   cost nothing (the bound of pre-landing N/32 of the next polynomial in LDS: k32 = all loads free)
#   st / st2: per-phase stamp timelines.  Then tools/power_model (package power / clock / voltage / throttle residencies).
O=gpurun_out/r05a
mkdir -p $O
smi() { for i in $(seq 1 $1); do sleep 0.4; rocm-smi --showpower --showclocks --csv 2>/dev/null | grep card0 | awk -F, '{print "      smi: sclk " $6 " power " $NF " W"}'; done; }
( rocm-smi -a > $O/smi_all.txt 2>&1; amd-smi metric > $O/amdsmi_metric.txt 2>&1; amd-smi static > $O/amdsmi_static.txt 2>&1 ) 
for p in 1 2; do
  for v in base g t x tx gtx c s k8 k16 k24 k32; do
    echo "== r5_$v (process $p) 8192 polynomials"
    smi 7 &
    KB_PAIR=1 KB_B2B=2 ./tools/kbench_r5_$v 8192 150 20 40 | grep -E "^pair|^forward|^inverse"
    wait
  done
done
for v in base g t x gtx c s k16 k32; do
  echo "== r5_$v 1024 polynomials"
  smi 5 &
  KB_PAIR=1 KB_B2B=8 ./tools/kbench_r5_$v 1024 300 20 300 | grep -E "^pair|^forward|^inverse"
  wait
done
for v in st st2; do
  echo "== r5_$v (stamped build: timeline) 8192 polynomials"
  KB_B2B=2 KB_WGDUMP=1 ./tools/kbench_r5_$v 8192 20 20 20
  echo "== r5_$v (stamped build: timeline) 1024 polynomials"
  KB_B2B=4 ./tools/kbench_r5_$v 1024 20 20 100
done
echo "== power model"
./tools/power_model 2.0 grid
