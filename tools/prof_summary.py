#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + per-kernel mean PMC values) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
print("== kernel stats ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print("%-90s calls=%s avg_ns=%s total_ns=%s pct=%s" % (row.get("Name", "")[:90], row.get("Calls"), row.get("AverageNs"),
                                                             row.get("TotalDurationNs"), row.get("Percentage")))
print("== counters (mean per dispatch) ==")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        short = name.replace("(anonymous namespace)::", "").split("(")[0][-60:]
        acc[short][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
for k in sorted(acc):
    if any(x in k for x in ("k_forward", "k_inverse", "k_polymul", "k_pointwise", "k_lat_", "k_ntt30", "tables_check")):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("    %-28s mean=%.4g  n=%d" % (c, sum(v) / len(v), len(v)))
