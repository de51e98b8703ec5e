#!/bin/bash
# usage: tools/build_kbench.sh <tag> [extra hipcc flags...]  -> tools/kbench_<tag>; prints VGPR / scratch figures of its kernels
# A variant build substitutes the tuning struct: -DMI355NTT_TUNE_HEADER='"/path/to/my_tune.hpp"' (a header that defines mi355ntt::Tune
# with the members of ntt-cuda_amd/csrc/tune.hpp); the kernel sources themselves carry no switches (round 6).
R=/root/repo
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $R/ntt-cuda_amd/csrc -I $R/include "$@" $R/tools/kbench.hip $R/ntt-cuda_amd/csrc/hostparams.cpp \
   -o $R/tools/kbench_$tag -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs:|VGPRs Spill|ScratchSize" | sed 's/.*remark: //; s/\[-Rpass.*//' | tr '\n' ' '
echo " <- $tag"
