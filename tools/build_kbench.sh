#!/bin/bash
# usage: tools/build_kbench.sh <tag> [extra hipcc flags...]  -> tools/kbench_<tag>; prints spill counts
R=/root/repo
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $R/ntt-cuda_amd/csrc -I $R/include "$@" $R/tools/kbench.hip $R/ntt-cuda_amd/csrc/hostparams.cpp \
   -o $R/tools/kbench_$tag -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs Spill" | sed 's/.*remark: //; s/\[-Rpass.*//; s/VGPRs Spill://' | tr '\n' ' '
echo " <- $tag (fwd hl6,4,2 | inv hl6,4,2)"
