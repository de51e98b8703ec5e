#!/bin/bash
# round 5, batch B: (1) kernel variants at 8192 and 1024 polynomials with the SMU's throttle residencies sampled (tools/smi_watch):
#   base; pl = inverse pre-landing (next polynomial's first column half by LDS-direct loads behind the exchange); pm3 / pm0 = memory
#   phase of both kernels at priority 3 / 0; plpm = pl + inverse memory phase at 3; ml = merged inverse row loads; n1s = shift-subtract
#   instead of one multiply-add (60-bit near-2^k); cs = canonicalisation by sign; mc0 = cross terms as four v_mul_lo_u32 + two v_add3
#   (2) the GPU parity suite on the library as built (here: -DMI355NTT_INV_PRELAND=1 and the round-5 BFV kernels)
#   (3) rocprofv3 kernel trace of the BFV drivers  (4) tools/power_model with chip-wide duty cycle
O=gpurun_out/r05b
mkdir -p $O
export TMPDIR=/tmp
BUS=$(python3 -c "import torch; print(torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0),'pci_bus_id') else '')" 2>/dev/null)
run() {   # tag size reps b2b warm
  ./tools/smi_watch 150 0 > $O/smi_$1_$2_$6.log 2>&1 &
  SMI=$!
  KB_PAIR=1 KB_B2B=$4 ./tools/kbench_r5_$1 $2 $3 20 $5 | grep -E "^pair|^forward|^inverse"
  kill $SMI; wait $SMI 2>/dev/null
  python3 tools/pviol.py $O/smi_$1_$2_$6.log 4
}
for p in 1 2; do
  for v in base pl pm3 pm0 plpm ml n1s cs mc0; do
    echo "== r5_$v (process $p) 8192 polynomials"
    run $v 8192 150 2 40 $p
  done
done
for p in 1 2; do
  for v in base pl pm3 plpm n1s mc0; do
    echo "== r5_$v (process $p) 1024 polynomials"
    run $v 1024 500 8 300 $p
  done
done
echo "== pytest -m gpu"
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5
echo "== BFV profile"
for set in 5 16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bfv$set/trace -- python3 tools/prof_driver_bfv.py 20 $set > $O/bfv$set.log 2>&1
  R=$set; [ $set = 5 ] && R=5
  python3 tools/prof_summary_bfv.py $O/bfv$set $R > $O/bfv${set}_summary.txt 2>&1
  cat $O/bfv${set}_summary.txt
done
echo "== power model (duty cycle in step)"
./tools/power_model 1.6 grid
