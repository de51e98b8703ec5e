#!/usr/bin/env python3
"""Latency path vs persistent path at n = 2^15 for small batch counts (set MI355NTT_LATENCY_PATH_MAX before import)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, ROOT)
import torch
import ntt_cuda_amd as ntt
from bench import Q60, PSI60, synth
dev = torch.device("cuda", 0)
ctx = ntt.NTTContext(32768, Q60, PSI60)
for num in [int(x) for x in os.environ.get("NUMS", "1,4,16,32,48,64,96,128,192,256,384,512").split(",")]:
    a = synth(torch, num, 32768, Q60, dev, 1)
    for _ in range(5):
        ctx.forward_batch(a, num); ctx.inverse_batch(a, num)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ctx.forward_batch(a, num); ctx.inverse_batch(a, num)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("num=%4d  pair %8.1f us  => %.3f M pairs/s" % (num, us, num / us))
