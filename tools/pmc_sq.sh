#!/bin/bash
# SQ-level counters for the single-pass kernels (one --pmc pass per group; no tracing domains).
# Output: gpurun_out/sq/  (summary printed)
set -u
OUT=gpurun_out/sq
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o -E "\b(SQ_[A-Z_0-9]+|SQC_[A-Z_0-9]+)\b" $OUT/avail.txt | sort -u > $OUT/sq_names.txt
wc -l $OUT/sq_names.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU" \
           "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 tools/prof_driver.py 1024 3 > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/sq/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "15" in k and "mi355ntt" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
tail -3 $OUT/g*.log | grep -i -E "error|invalid|not" | head
