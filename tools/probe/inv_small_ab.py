#!/usr/bin/env python3
"""probe (round 5): inverse launches and forward -> inverse pairs on the smaller rings at 256 MiB / 1 GiB / 2 GiB, shipped library
against another build (MI355NTT_LIB); ring degree from MI355NTT_PROBE_N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ntt_cuda_amd as ntt, params as P
if os.environ.get("MI355NTT_LIB"):
    ntt.LIB_PATH = os.environ["MI355NTT_LIB"]
dev = torch.device("cuda", 0)
n = int(os.environ.get("MI355NTT_PROBE_N", "4096"))
psis = [pow(p, 32768 // n, q) for p, q in zip(P.PSI60, P.Q60)]
ctx = ntt.NTTContext(n, P.Q60, psis)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = []
for mib, reps in ((256, 300), (1024, 80), (2048, 40)):
    num = mib * (1 << 20) // (8 * n)
    a = torch.empty((num, n), dtype=torch.int64, device=dev); ctx.synth_splitmix(a, num, 5)
    def rate(fn):
        for _ in range(reps): fn()
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    p = rate(lambda: (ctx.forward_batch(a, num), ctx.inverse_batch(a, num)))
    i = rate(lambda: ctx.inverse_batch(a, num))
    out.append("%4d MiB: pair %.4f  inv %.4f" % (mib, p, i))
print("# n = %d  lib = %s :  %s" % (n, os.path.basename(os.environ.get("MI355NTT_LIB") or "shipped"), "   ".join(out)))
