#!/usr/bin/env python3
"""30-bit path, n = 65536: forward / inverse time per call against the batch size (run with and without MI355NTT_NO_PAIR16=1:
pair launches from kPair30MinPolys polynomials, or the stage launch + two half-size transforms everywhere)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import ntt_cuda_amd as ntt
from bench30 import setup30
dev = torch.device("cuda", 0)
n = 65536
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("mode:", "no pair launches" if os.environ.get("MI355NTT_NO_PAIR16") else "default")
for num in (16, 32, 48, 64, 96, 128, 192, 256, 512, 1024):
    a, q, mu, bits, tab, tabi = setup30(torch, ntt, n, num, dev)
    res = []
    for fn in (lambda: ntt.forward30(a, n, q, mu, bits, tab, num), lambda: ntt.inverse30(a, n, q, mu, bits, tabi, num)):
        for _ in range(60): fn()
        e0.record()
        for _ in range(60): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 60 * 1e3)
    print("num=%5d  fwd %7.1f us  inv %7.1f us" % (num, res[0], res[1]))
