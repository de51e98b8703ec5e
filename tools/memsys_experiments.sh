#!/bin/bash
# What bounds k_forward15 / k_inverse15 besides VALU issue: timing experiments with tools/kbench.hip (results of the
# ablated builds are wrong by construction, timing only).
#   tools/memsys_experiments.sh build     (here: hipcc cross-compiles)      -> tools/kbench_<tag>
#   tools/memsys_experiments.sh run       (on the GPU box)                  -> gpurun_out/memsys.txt
set -u
cd "$(dirname "$0")/.."
if [ "${1:-run}" = build ]; then
  tools/build_kbench.sh base   -DMI355NTT_STAGGER_FWD=0 &
  tools/build_kbench.sh ntld   -DMI355NTT_STAGGER_FWD=0 -DMI355NTT_STREAM_AUX_LD=2 &
  tools/build_kbench.sh ntst   -DMI355NTT_STAGGER_FWD=0 -DMI355NTT_STREAM_AUX_ST=2 &
  tools/build_kbench.sh ship &
  wait
  tools/build_kbench.sh sginv  -DMI355NTT_STAGGER_INV=1 &
  tools/build_kbench.sh sg4    -DMI355NTT_STAGGER_FWD=4 &
  tools/build_kbench.sh l2     '-DMI355NTT_POLY_SLOT(y)=((y)%8u)' &
  tools/build_kbench.sh mall   '-DMI355NTT_POLY_SLOT(y)=((y)%256u)' &
  wait
  tools/build_kbench.sh twl1   -DMI355NTT_ABLATE_TWL1 &
  tools/build_kbench.sh notw   -DMI355NTT_ABLATE_TWIDDLE &
  tools/build_kbench.sh noex   -DMI355NTT_ABLATE_EXCHANGE &
  tools/build_kbench.sh st     -DMI355NTT_STAMPS &
  wait
  tools/build_kbench.sh stl2   -DMI355NTT_STAMPS '-DMI355NTT_POLY_SLOT(y)=((y)%8u)'
  exit 0
fi
OUT=gpurun_out/memsys.txt
mkdir -p gpurun_out
{
echo "# kbench <num> <reps> <hl: 20 = 60-bit class, all primes near 2^k> <warm launches>; median of reps, warm clocks"
echo "# base = no stagger; ntld / ntst = base with non-temporal loads / stores; ship = the shipped kernels (forward stagger 1); sg4 / sginv = stagger 4 / stagger also on the inverse"
for n in 256 1024 4096; do
  for v in base ntld ntst ship sg4 sginv; do echo "== $v num=$n"; ./tools/kbench_$v $n 20 20 300 | grep -E "forward|inverse"; done
done
echo "# polynomial index folded so the batch stays resident: l2 = 8 slots (2 MiB), mall = 256 slots (64 MiB)"
for v in ship mall l2; do echo "== $v num=1024"; ./tools/kbench_$v 1024 20 20 300 | grep -E "forward|inverse"; done
echo "# twiddle loads: all inside one 4 KiB window (L1 hits) / none (synthesised) / no workgroup exchange"
for v in twl1 notw noex; do echo "== $v num=1024"; ./tools/kbench_$v 1024 20 20 300 | grep -E "forward|inverse"; done
echo "# per-phase timeline of k_forward15, HBM vs L2-resident data"
for v in st stl2; do echo "== $v num=1024"; ./tools/kbench_$v 1024 20 20 400 | grep -v inverse; done
} > $OUT 2>&1
tail -5 $OUT
