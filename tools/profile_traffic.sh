#!/bin/bash
# HBM traffic per launch of the dominant kernels from PMC counters (separate --pmc passes, as the guide
# prescribes), plus the calibration copies.  Output: gpurun_out/traffic/
set -u
OUT=gpurun_out/traffic
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/calib_$c -- ./tools/calib_copy > $OUT/calib_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $OUT/ntt_$c -- python3 tools/prof_driver.py 1024 3 > $OUT/ntt_$c.log 2>&1
done
python3 tools/traffic_summary.py $OUT
