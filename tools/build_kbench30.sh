#!/bin/bash
# usage: tools/build_kbench30.sh <tag> [extra hipcc flags...]  -> tools/kbench30_<tag>; prints register use of the n = 2^15 kernels
R=/root/repo
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $R/ntt-cuda_amd/csrc -I $R/include "$@" $R/tools/kbench30.hip \
   -o $R/tools/kbench30_$tag -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill" | sed 's/.*remark: //; s/\[-Rpass.*//' | paste - - - | grep -E "error|ntt30xILi15" | sed 's/Function Name: _ZN8mi355ntt12_GLOBAL__N_1//'
echo " <- $tag"
