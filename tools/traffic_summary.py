#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes into bytes per launch, calibrated on known-size copies."""
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
def mean_by_kernel(pattern):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, pattern, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(float(row["Counter_Value"]))
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

# the committed form (profiles/traffic_rNN.json, read by bench.py): bytes per launch with the gfx950 FETCH_SIZE correction
# the calibration copies establish (x 2), against the algorithmic bytes of one launch of the profiled workload
ALG = {"k_forward15<": 1024 * 524288, "k_inverse15<": 1024 * 524288, "k_polymul15<": 1024 * 786432,
       "k_forward15_lit": 1024 * 524288, "k_inverse15_lit": 1024 * 524288,
       "k_ntt30x<15, true>": 1024 * 262144, "k_ntt30x<15, false>": 1024 * 262144}
fs = [c["fetch_scale"] for c in res["calibration"].values()]
scale = round(sum(fs) / len(fs)) if fs else 2
final = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_traffic.sh), python3 tools/prof_driver.py "
                 "1024 3: n=32768, 4x60-bit primes, 1024 polynomials per launch (30-bit kernels: 1024 polynomials of 32768 words). "
                 "rocprofv3 reports KiB. Calibration on known 512 MiB copies (tools/calib_copy.hip) on the same box gives the FETCH_SIZE "
                 "scale (%d: FETCH_SIZE reads half of the bytes on gfx950, as MI355X_MICROARCH.md says), WRITE_SIZE is exact. "
                 "hbm_bytes_per_launch = (%d*FETCH_SIZE + WRITE_SIZE)*1024; Infinity-Cache hits are included in FETCH_SIZE." % (scale, scale),
         "calibration": res["calibration"]}
for k, v in res["kernels"].items():
    for name, alg in ALG.items():
        if ("::" + name) in k and v.get("write_KiB_raw") is not None:
            hbm = (scale * v["fetch_KiB_raw"] + v["write_KiB_raw"]) * 1024
            final[name.replace("<15, true>", "_fwd15").replace("<15, false>", "_inv15").rstrip("<")] = dict(
                v, hbm_bytes_per_launch=hbm, algorithmic_bytes_per_launch=alg, ratio=hbm / alg)
# what the figures belong to: the instruction streams of the profiled kernels in the library that ran (tools/codeobj_digest.py) and the
# commit of the tree -- tests/test_abi_host.py fails when the shipped kernels are no longer these, i.e. when the passes must be re-run
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
try:
    import subprocess
    import codeobj_digest
    dig = codeobj_digest.named_digests(os.path.join(ROOT, "ntt-cuda_amd", "build", "kernels_fast_n15.hip.o"))
    final["kernel_digest"] = {k: v for k, v in dig.items() if k.startswith(("k_forward15<4, true, 0", "k_inverse15<4, true", "k_polymul15<4, true", "k_forward15_lit", "k_inverse15_lit"))}
    final["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("MI355NTT_COMMIT", "")
except Exception as exc:
    final["kernel_digest"] = {"error": repr(exc)}
json.dump(final, open(os.path.join(out, "traffic_final.json"), "w"), indent=1)
