#!/usr/bin/env python3
"""Small driver for rocprofv3: a few forward/inverse/polymul launches at the bench workload."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd"))
sys.path.insert(0, ROOT)
import torch
import ntt_cuda_amd as ntt
from bench import Q60, PSI60, synth

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
ctx = ntt.NTTContext(32768, Q60, PSI60)
a = synth(torch, batch, 32768, Q60, dev, 1)
b = synth(torch, batch, 32768, Q60, dev, 2)
# the same shape as bench.py: enough back-to-back launches for the clocks to settle (a handful of cold launches reads
# ~10 % slow), so the per-kernel averages of `rocprofv3 --kernel-trace --stats` are comparable with bench.py's
for _ in range(reps):
    ctx.forward_batch(a, batch)
    ctx.inverse_batch(a, batch)
ctx.forward_batch(b, batch)
for _ in range(max(1, reps // 4)):
    ctx.polymul_batch(a, b, batch)
torch.cuda.synchronize()
print("done")
