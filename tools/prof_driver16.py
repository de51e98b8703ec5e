#!/usr/bin/env python3
"""Small driver for rocprofv3: forward / inverse / fused product launches at n = 2^16, 512 polynomials (256 MiB), one 60-bit prime.
MI355NTT_NO_PAIR16=1 in the environment: the single-workgroup forward launch instead of the pair launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd"))
sys.path.insert(0, ROOT)
import torch
import ntt_cuda_amd as ntt
from bench import Q60, synth

n, num = 65536, 512
q = Q60[0]
psi = next(pw for pw in (pow(x, (q - 1) // (2 * n), q) for x in range(2, 1000)) if pow(pw, n, q) == q - 1)
dev = torch.device("cuda", 0)
ctx = ntt.NTTContext(n, [q], [psi])
a, b = synth(torch, num, n, [q], dev, 1), synth(torch, num, n, [q], dev, 2)
ctx.forward_batch(b, num)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    ctx.forward_batch(a, num)
    ctx.inverse_batch(a, num)
for _ in range(3):
    ctx.polymul_batch(a, b, num)
torch.cuda.synchronize()
print("done")
