#!/bin/bash
# rocprofv3 kernel trace of the n = 2^16 launches (pair forward / fused inverse / fused product, then the single-workgroup
# forward), 512 polynomials.  Output: gpurun_out/prof16/
set -u
OUT=gpurun_out/prof16
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_driver16.py 100 > $OUT/trace.log 2>&1
MI355NTT_NO_PAIR16=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nopair -- python3 tools/prof_driver16.py 100 > $OUT/trace_nopair.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for tag in ("trace", "trace_nopair"):
    print("== %s ==" % ("default dispatch (pair forward)" if tag == "trace" else "MI355NTT_NO_PAIR16=1 (single-workgroup forward)"))
    for f in glob.glob(os.path.join(out, tag, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "mi355ntt::k_" in row.get("Name", ""):
                print("%-100s calls=%s avg_ns=%s" % (row["Name"][:100], row.get("Calls"), row.get("AverageNs")))
PY
