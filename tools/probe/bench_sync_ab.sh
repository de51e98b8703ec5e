#!/bin/bash
# A/B of bench.py's bracket on ONE box: polling in front of barrier + synchronize (default) against --blocking-sync, the driver's K / W
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for i in 1 2 3 4; do
  for m in "" "--blocking-sync"; do
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $m 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-16s value %.4f M  wall %.4f ms  events %.4f ms  rounds median %.4f M  settled %s' % ('$m' or 'polling', d['value']/1e6, d['ms_per_step'], d['kernel_ms']['step_by_events'], d['rounds']['pairs_per_s_median']/1e6, d['settled']))
"
  done
done | tee gpurun_out/bench_sync_ab.txt
