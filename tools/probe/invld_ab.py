#!/usr/bin/env python3
"""probe (round 5): the shipped library against a build with k_inverse15's row loads at the default cache policy
(MI355NTT_LIB=<path>): the bench's own step (forward_batch, inverse_batch alternating), each kernel on its own, the fused product"""
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
for num, reps in ((1024, 300), (2048, 150), (4096, 80), (8192, 40), (640, 300)):
    a = torch.empty((num, n), dtype=torch.int64, device=dev); ctx.synth_splitmix(a, num, 5)
    b = a.clone(); ctx.forward_batch(b, num)
    def rate(fn):
        for _ in range(reps): fn()
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    p = rate(lambda: (ctx.forward_batch(a, num), ctx.inverse_batch(a, num)))
    f = rate(lambda: ctx.forward_batch(a, num))
    i = rate(lambda: ctx.inverse_batch(a, num))
    m = rate(lambda: ctx.polymul_batch(a, b, num))
    out.append("%5d: pair %.4f ms (%.3f M/s)  fwd %.4f  inv %.4f  fused %.4f" % (num, p, num / p / 1e3, f, i, m))
print("# lib = %s" % (os.environ.get("MI355NTT_LIB") or "shipped"))
print("\n".join(out))
