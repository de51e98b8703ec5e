#!/bin/bash
# Round-2 experiment batch A: wave-priority schemes, merged inverse loads, workgroup timeline.
#   tools/exp_r02a.sh build   (here)      tools/exp_r02a.sh run   (GPU box) -> gpurun_out/exp_r02a.txt
set -u
cd "$(dirname "$0")/.."
B=tools/build_kbench.sh
if [ "${1:-run}" = build ]; then
  $B ship & $B st -DMI355NTT_STAMPS & $B im -DMI355NTT_INV_MERGED_LOADS=1 & $B sg0 -DMI355NTT_STAGGER_FWD=0 &
  wait
  for s in 8 9 10 11 12 13; do $B dyn$s -DMI355NTT_DYNPRIO=$s & done
  wait
  # static descending priorities, forward: R2 3 | R3 2 | R1 first 3 stages 1 | R1 rest 0
  $B fA -DMI355NTT_PRIO_R1=1 -DMI355NTT_PRIO_R2=3 -DMI355NTT_PRIO_R3=2 -DMI355NTT_PSPLIT_R1=12 -DMI355NTT_PRIO_R1B=0 &
  # R2 3 | R3 first 2 stages 2, rest 1 | R1 0
  $B fB -DMI355NTT_PRIO_R1=0 -DMI355NTT_PRIO_R2=3 -DMI355NTT_PRIO_R3=2 -DMI355NTT_PSPLIT_R3=8 -DMI355NTT_PRIO_R3B=1 &
  # R2 3 | R3 2 -> 1 after 3 stages | R1 1 -> 0 after 3 stages
  $B fC -DMI355NTT_PRIO_R1=1 -DMI355NTT_PRIO_R2=3 -DMI355NTT_PRIO_R3=2 -DMI355NTT_PSPLIT_R3=12 -DMI355NTT_PRIO_R3B=1 -DMI355NTT_PSPLIT_R1=12 -DMI355NTT_PRIO_R1B=0 &
  # R2 3 -> 2 after 3 stages | R3 2 -> 1 after 2 stages | R1 1 -> 0 after 2 stages
  $B fD -DMI355NTT_PRIO_R1=1 -DMI355NTT_PRIO_R2=3 -DMI355NTT_PSPLIT_R2=12 -DMI355NTT_PRIO_R2B=2 -DMI355NTT_PRIO_R3=2 -DMI355NTT_PSPLIT_R3=8 -DMI355NTT_PRIO_R3B=1 -DMI355NTT_PSPLIT_R1=8 -DMI355NTT_PRIO_R1B=0 &
  # no priorities at all
  $B p0 -DMI355NTT_PRIO_R1=0 -DMI355NTT_PRIO_R2=0 -DMI355NTT_PRIO_R3=0 -DMI355NTT_PRIO_I1=0 -DMI355NTT_PRIO_I2=0 -DMI355NTT_PRIO_I3=0 &
  # inverse, descending from the exchange: R3' 3 | R1' 2 | R2' 1 -> 0 after 2 stages   (+ merged loads)
  $B iA -DMI355NTT_INV_MERGED_LOADS=1 -DMI355NTT_PRIO_I3=3 -DMI355NTT_PRIO_I1=2 -DMI355NTT_PRIO_I2=1 -DMI355NTT_PSPLIT_I2=8 -DMI355NTT_PRIO_I2B=0 &
  # R3' 3 | R1' 2 -> 1 after 3 stages | R2' 1 -> 0 after 3 stages
  $B iB -DMI355NTT_INV_MERGED_LOADS=1 -DMI355NTT_PRIO_I3=3 -DMI355NTT_PRIO_I1=2 -DMI355NTT_PSPLIT_I1=12 -DMI355NTT_PRIO_I1B=1 -DMI355NTT_PRIO_I2=1 -DMI355NTT_PSPLIT_I2=12 -DMI355NTT_PRIO_I2B=0 &
  wait
  $B dyn10im -DMI355NTT_DYNPRIO=10 -DMI355NTT_INV_MERGED_LOADS=1 &
  $B stdyn10 -DMI355NTT_STAMPS -DMI355NTT_DYNPRIO=10 &
  wait
  exit 0
fi
OUT=gpurun_out/exp_r02a.txt
mkdir -p gpurun_out
{
./tools/kbench_ship 1024 5 20 600 > /dev/null      # settle the clocks
for rep in 1 2; do
for v in ship sg0 im p0 fA fB fC fD iA iB dyn8 dyn9 dyn10 dyn11 dyn12 dyn13 dyn10im; do
  echo "== $v num=1024 (pass $rep)"; ./tools/kbench_$v 1024 30 20 300 | grep -E "forward|inverse"
done
done
for v in ship dyn10 fC iA; do echo "== $v num=4096"; ./tools/kbench_$v 4096 10 20 100 | grep -E "forward|inverse"; done
for v in ship dyn10 fC iA; do echo "== $v num=256"; ./tools/kbench_$v 256 30 20 600 | grep -E "forward|inverse"; done
for v in st stdyn10; do echo "== $v num=1024"; ./tools/kbench_$v 1024 20 20 400; done
} > $OUT 2>&1
tail -3 $OUT
