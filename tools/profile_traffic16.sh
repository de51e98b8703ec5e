#!/bin/bash
# HBM traffic per launch of the n = 2^16 kernels (pair / single-workgroup forward, fused inverse) from PMC counters, separate
# --pmc passes, with the calibration copies of tools/profile_traffic.sh.  Output: gpurun_out/traffic16/
set -u
OUT=gpurun_out/traffic16
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/calib_$c -- ./tools/calib_copy > $OUT/calib_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $OUT/pair_$c -- python3 tools/prof_driver16.py 4 > $OUT/pair_$c.log 2>&1
  MI355NTT_NO_PAIR16=1 rocprofv3 --pmc $c --output-format csv -d $OUT/nopair_$c -- python3 tools/prof_driver16.py 4 > $OUT/nopair_$c.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
def mean_by_kernel(pattern):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, pattern, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
BYTES = 512 << 20
cf, cw = mean_by_kernel("calib_FETCH_SIZE"), mean_by_kernel("calib_WRITE_SIZE")
fs = [BYTES / (cf[k][0] * 1024) for k in cf if "copy" in k]
scale = round(sum(fs) / len(fs)) if fs else 2
res = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_traffic16.sh), python3 tools/prof_driver16.py 4: "
               "n = 65536, one 60-bit prime, 512 polynomials per launch = 268 435 456 B read + as many written algorithmically (fused product: + 268 435 456 B "
               "of the second operand).  rocprofv3 reports KiB; FETCH_SIZE scale %d from the 512 MiB calibration copies of the same box (gfx950 counts half), "
               "WRITE_SIZE exact.  FETCH_SIZE counts what leaves the L2s (Infinity-Cache hits included)." % scale,
       "fetch_scale": scale, "kernels": {}}
for tag in ("pair", "nopair"):
    nf, nw = mean_by_kernel(tag + "_FETCH_SIZE"), mean_by_kernel(tag + "_WRITE_SIZE")
    for k in nf:
        if "mi355ntt::k_" in k and ("15" in k):
            name = k.strip().replace("mi355ntt::", "")
            rd, wr = scale * nf[k][0] * 1024, nw.get(k, (0, 0))[0] * 1024
            res["kernels"]["%s [%s run, %d launches]" % (name, tag, nf[k][1])] = {
                "read_bytes_per_launch": rd, "written_bytes_per_launch": wr, "read_over_algorithmic": rd / (268435456.0), "written_over_algorithmic": wr / 268435456.0}
json.dump(res, open(os.path.join(out, "traffic16.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
