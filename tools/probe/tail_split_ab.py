#!/usr/bin/env python3
"""probe: batches of k x 256 + r polynomials at n = 2^15 with and without the tail split (MI355NTT_NO_TAIL_SPLIT=1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ntt_cuda_amd as ntt, params as P
dev = torch.device("cuda", 0)
n = 32768
ctx = ntt.NTTContext(n, P.Q60, P.PSI60)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("# NO_TAIL_SPLIT=%s   num: fwd us, inv us, pair us, fused product us" % os.environ.get("MI355NTT_NO_TAIL_SPLIT"))
for num in (288, 320, 352, 384, 400, 416, 512 + 32, 512 + 64, 512 + 96, 640, 512 + 144, 512 + 160, 512 + 176, 768 + 64, 1024 + 64, 1024 + 128, 2048 + 100):
    a = torch.empty((num, n), dtype=torch.int64, device=dev); ctx.synth_splitmix(a, num, 5)
    b = a.clone(); ctx.forward_batch(b, num)
    def rate(fn):
        for _ in range(30): fn()
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 30 * 1e3
    f = rate(lambda: ctx.forward_batch(a, num))
    i = rate(lambda: ctx.inverse_batch(a, num))
    p = rate(lambda: (ctx.forward_batch(a, num), ctx.inverse_batch(a, num)))
    m = rate(lambda: ctx.polymul_batch(a, b, num))
    print("%5d: %7.1f %7.1f %7.1f %7.1f" % (num, f, i, p, m))
