#!/usr/bin/env python3
"""Latency path vs single-pass path for small batch counts (set MI355NTT_LATENCY_PATH_MAX before import; N=<ring degree>,
default 32768; NUMS=comma list).  Prints forward+inverse pair time and fused-product time per batch size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, ROOT)
import torch
import ntt_cuda_amd as ntt
from bench import Q60
n = int(os.environ.get("N", "32768"))
psis = [next(x for x in (pow(g, (q - 1) // (2 * n), q) for g in range(2, 500)) if pow(x, n, q) == q - 1) for q in Q60]
ctx = ntt.NTTContext(n, Q60, psis)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(f, reps=50):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for num in [int(x) for x in os.environ.get("NUMS", "1,4,16,32,48,64,96,128,192,256,384,512").split(",")]:
    a = torch.empty((num, n), dtype=torch.int64, device="cuda:0"); b = torch.empty_like(a)
    ctx.synth_splitmix(a, num, 1); ctx.synth_splitmix(b, num, 777)
    def pair():
        ctx.forward_batch(a, num); ctx.inverse_batch(a, num)
    us, um = t(pair), t(lambda: ctx.polymul_batch(a, b, num))
    print("n=%6d num=%5d  pair %8.1f us  => %.3f M pairs/s   polymul %8.1f us => %.3f M/s" % (n, num, us, num / us, um, num / um))
