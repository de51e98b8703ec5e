#!/usr/bin/env python3
"""rocprofv3 driver: the small-batch (two-launch) kernels forced onto larger batches (MI355NTT_LATENCY_PATH_MAX in the environment),
n = 2^15, 4 x 60-bit primes: how long do k_lat_* take when the batch fills the chip -- from HBM (512, 1024 polynomials) and from the
Infinity Cache (128 polynomials = 32 MiB)?  `python3 tools/prof_driver_lat.py summary <dir>` prints the per-batch means of a trace."""
import os, sys
if len(sys.argv) > 2 and sys.argv[1] == "summary":
    import csv, glob
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seg, acc, order = 0, {}, []
    for r in rows:
        n = r["Kernel_Name"]
        if "pointwise" in n:
            seg += 1
            continue
        if "k_lat_" not in n:
            continue
        k = (seg, n.split("<")[0].split("::")[-1])
        acc.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    nums = (128, 256, 512, 1024)
    for k in sorted(acc):
        v = acc[k][5:]
        print("polynomials=%4d  %-12s launches=%d  mean %.1f us" % (nums[k[0]] if k[0] < len(nums) else -1, k[1], len(v), sum(v) / len(v) / 1000))
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, ROOT)
import torch
import ntt_cuda_amd as ntt
from bench import Q60, PSI60, synth
dev = torch.device("cuda", 0)
ctx = ntt.NTTContext(32768, Q60, PSI60)
for num in (128, 256, 512, 1024):
    a = synth(torch, num, 32768, Q60, dev, 1)
    for _ in range(30):
        ctx.forward_batch(a, num)
        ctx.inverse_batch(a, num)
    torch.cuda.synchronize()
    ctx.pointwise_mul(a, a, a, num)      # (marker between the batches in the trace)
torch.cuda.synchronize()
print("done")
