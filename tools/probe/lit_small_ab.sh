#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
MI355NTT_LATENCY_PATH_MAX=1000000 python -m pytest tests/test_gpu_round6.py tests/test_gpu_fuzz_moduli.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -4
MI355NTT_LATENCY_PATH_MAX=0 python tools/probe/lit_small_ab.py 2>&1 | grep path > /tmp/p0.txt
MI355NTT_LATENCY_PATH_MAX=1000000 python tools/probe/lit_small_ab.py 2>&1 | grep path > /tmp/p1.txt
paste -d'|' /tmp/p0.txt /tmp/p1.txt | awk -F'|' '{print $1; print $2; print ""}' | head -400
} 2>&1 | tee gpurun_out/lit_small_ab.txt
