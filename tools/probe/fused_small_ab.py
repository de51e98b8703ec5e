#!/usr/bin/env python3
"""probe (round 5): the fused product at cache-resident batch sizes, shipped library against another build (MI355NTT_LIB)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ntt_cuda_amd as ntt, params as P
if os.environ.get("MI355NTT_LIB"):
    ntt.LIB_PATH = os.environ["MI355NTT_LIB"]
dev = torch.device("cuda", 0)
n = 32768
ctx = ntt.NTTContext(n, P.Q60, P.PSI60)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = []
for num in [int(x) for x in os.environ.get("MI355NTT_PROBE_SIZES", "100,128,192,256,320,384,512,768").split(",")]:
    a = torch.empty((num, n), dtype=torch.int64, device=dev); ctx.synth_splitmix(a, num, 5)
    b = a.clone(); ctx.forward_batch(b, num)
    def rate(fn, reps=400):
        for _ in range(reps): fn()
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    out.append("%4d: %.1f" % (num, rate(lambda: ctx.polymul_batch(a, b, num))))
print("# lib = %s   fused product, us per launch:  %s" % (os.path.basename(os.environ.get("MI355NTT_LIB") or "shipped"), "   ".join(out)))
