#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes into bytes per launch, calibrated on known-size copies."""
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
def mean_by_kernel(pattern):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, pattern, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
BYTES = 512 << 20
res = {"units": "FETCH_SIZE/WRITE_SIZE are reported in KiB by rocprofv3", "calibration": {}, "kernels": {}}
cf, cw = mean_by_kernel("calib_FETCH_SIZE"), mean_by_kernel("calib_WRITE_SIZE")
for k in cf:
    if "copy" in k:
        res["calibration"][k.strip()] = {"fetch_KiB": cf[k], "write_KiB": cw.get(k), "true_bytes_each_way": BYTES,
                                         "fetch_scale": BYTES / (cf[k] * 1024), "write_scale": BYTES / (cw.get(k, 1) * 1024)}
nf, nw = mean_by_kernel("ntt_FETCH_SIZE"), mean_by_kernel("ntt_WRITE_SIZE")
for k in nf:
    if "mi355ntt" in k:
        res["kernels"][k.strip()] = {"fetch_KiB_raw": nf[k], "write_KiB_raw": nw.get(k)}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, "traffic_raw.json"), "w"), indent=1)
