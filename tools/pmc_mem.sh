#!/bin/bash
# L1/L2-level counters for the single-pass kernels (one --pmc pass per group; no tracing domains).
set -u
OUT=gpurun_out/mem
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" \
           "TCC_STREAMING_REQ_sum TCC_NORMAL_EVICT_sum TCC_NORMAL_WRITEBACK_sum" \
           "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 tools/prof_driver.py 1024 3 > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/mem/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "15" in k and "mi355ntt" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-36s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
