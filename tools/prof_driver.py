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
for _ in range(max(6, reps // 4)):
    ctx.polymul_batch(a, b, batch)
# kernel class 0 (round 6): a context on a Barrett-inexact 60-bit modulus -- the reference's own butterflies in the single-pass kernels
from bench import Q60_INEXACT, PSI60_INEXACT
ctx0 = ntt.NTTContext(32768, [Q60_INEXACT], [PSI60_INEXACT])
a0 = synth(torch, batch, 32768, [Q60_INEXACT], dev, 3)
for _ in range(max(6, reps // 2)):
    ctx0.forward_batch(a0, batch)
    ctx0.inverse_batch(a0, batch)
ctx0.close()
del a0
# the latency path (one polynomial: k_lat_fwd_a + k_lat_fwd_b, k_lat_inv_b + k_lat_inv_a, and the three-launch product)
one, two = a[:32768].clone(), b[:32768].clone()
for _ in range(max(6, reps // 4)):
    ctx.forward_batch(one, 1, 1)
    ctx.inverse_batch(one, 1, 1)
    ctx.polymul_batch(one, two, 1, 1)
# the reference-signature entry points (checked mode: table check + guarded fast kernels), same workload
import numpy as np
tabs_f = torch.empty((4, 32768), dtype=torch.int64, device=dev)
tabs_i = torch.empty((4, 32768), dtype=torch.int64, device=dev)
for i in range(4):
    tp, ti = ntt.fillTablePsi128(PSI60[i], Q60[i], ntt.modinv128(PSI60[i], Q60[i]), 32768)
    tabs_f[i] = torch.from_numpy(tp.view(np.int64))
    tabs_i[i] = torch.from_numpy(ti.view(np.int64))
mod = ntt.Moduli(Q60)
for _ in range(max(6, reps // 8)):
    ntt.forwardNTT_batch(a, 32768, tabs_f, batch, 4, mod)
    ntt.inverseNTT_batch(a, 32768, tabs_i, batch, 4, mod)
# the 30-bit path (old/ntt_30bit.cuh) on its native kernels, same ring size
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench30 import setup30
a30, q30, mu30, bits30, tf30, ti30 = setup30(torch, ntt, 32768, batch, dev)
for _ in range(max(6, reps // 8)):
    ntt.forward30(a30, 32768, q30, mu30, bits30, tf30, num=batch)
    ntt.inverse30(a30, 32768, q30, mu30, bits30, ti30, num=batch)
torch.cuda.synchronize()
print("done")
