#!/bin/bash
# round 5, batch F: (1) the batched BFV drivers under rocprofv3 with the round-4 element-wise kernels (variant library: bfv_host.cpp +
# kernels_bfv.hip of commit 88dd3cd linked against the round-5 objects) and with the shipped ones; (2) the inverse kernel's stamped
# timeline at 8192 polynomials with pre-landing.
O=gpurun_out/r05f
mkdir -p $O
export TMPDIR=/tmp
for lib in old new; do
  for set in 5 16; do
    if [ $lib = old ]; then export MI355NTT_LIB=$PWD/ntt-cuda_amd/build/libvar_oldbfv.so; else unset MI355NTT_LIB; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/bfv_${lib}_$set/trace -- python3 tools/prof_driver_bfv.py 20 $set > $O/bfv_${lib}_$set.log 2>&1
    python3 tools/prof_summary_bfv.py $O/bfv_${lib}_$set $set > $O/bfv_${lib}_${set}_summary.txt 2>&1
    echo "== BFV profile: $lib element-wise kernels, $set primes"
    grep -v "at::native\|rocclr\|k_lat_\|k_sample\|k_salsa\|k_keygen\|k_add" $O/bfv_${lib}_${set}_summary.txt
  done
done
unset MI355NTT_LIB
echo "== stamped timeline with pre-landing, 8192 polynomials"
KB_B2B=2 ./tools/kbench_r5_plst 8192 20 20 20 | grep -v "xcd \|phase "
