#!/bin/bash
# usage: tools/build_kbench.sh <tag> [extra hipcc flags...]  -> tools/kbench_<tag>; prints spill counts
# (only the <HL 4, near-2^k> instantiation of the n = 2^15 kernels is compiled: -DMI355NTT_ONLY_HL4N)
R=/root/repo
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $R/ntt-cuda_amd/csrc -I $R/include -DMI355NTT_LAB -DMI355NTT_ONLY_HL4N "$@" $R/tools/kbench.hip $R/ntt-cuda_amd/csrc/hostparams.cpp \
   -o $R/tools/kbench_$tag -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs:|VGPRs Spill|ScratchSize" | sed 's/.*remark: //; s/\[-Rpass.*//' | tr '\n' ' '
echo " <- $tag"
