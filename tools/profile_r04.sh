#!/bin/bash
# round 4 profile set (run on the GPU box from the repo root): rocprofv3 kernel trace + PMC passes at the bench batch (tools/profile.sh),
# a kernel trace at 4096 polynomials (the HBM-streaming state), the HBM traffic passes (tools/profile_traffic.sh), and the kernel trace of
# bench.py itself.  Counters are collected in their own passes (never with trace options gpurun refuses).
set -u
export TMPDIR=/tmp
bash tools/profile.sh 1024 > /dev/null 2>&1
cp gpurun_out/prof/summary.txt gpurun_out/prof/summary_batch1024.txt
OUT=gpurun_out/prof4096
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_driver.py 4096 60 > $OUT/trace.log 2>&1
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
mkdir -p gpurun_out/traffic; bash tools/profile_traffic.sh > gpurun_out/traffic/summary.txt 2>&1
OUT=gpurun_out/profbench
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-extras > $OUT/bench_line.json 2> $OUT/trace.log
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
