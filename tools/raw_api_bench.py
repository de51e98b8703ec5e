#!/usr/bin/env python3
"""Context API vs the reference-signature raw API (checked / trusted routing, literal kernels) on the bench workload."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ntt_cuda_amd as ntt
from bench import Q60, PSI60, synth

n, P, batch = 32768, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
ctx = ntt.NTTContext(n, Q60, PSI60)
a = synth(torch, batch, n, Q60, dev, 1)
a0 = a.clone()
tabs_f = torch.empty((P, n), dtype=torch.int64, device=dev)
tabs_i = torch.empty((P, n), dtype=torch.int64, device=dev)
for i in range(P):
    tp, ti = ntt.fillTablePsi128(PSI60[i], Q60[i], ntt.modinv128(PSI60[i], Q60[i]), n)
    tabs_f[i] = torch.from_numpy(tp.view(np.int64))
    tabs_i[i] = torch.from_numpy(ti.view(np.int64))
mod = ntt.Moduli(Q60)
tabs_f2, tabs_i2 = tabs_f.clone(), tabs_i.clone()           # a second pair of tables: one stays checked, one gets trusted
assert ntt.raw_uses_fast_kernels(n, tabs_f, mod) and ntt.raw_uses_fast_kernels(n, tabs_i, mod, inverse=True)
assert ntt.raw_trust_tables(n, tabs_f2, mod) and ntt.raw_trust_tables(n, tabs_i2, mod, inverse=True)
mu_lit = mod.mu.copy(); mu_lit[0] -= 1
lit = ntt.Moduli(Q60, mu=mu_lit, bits=mod.bits)


def ctx_step():
    ctx.forward_batch(a, batch); ctx.inverse_batch(a, batch)


def checked_step():
    ntt.forwardNTT_batch(a, n, tabs_f, batch, P, mod); ntt.inverseNTT_batch(a, n, tabs_i, batch, P, mod)


def trusted_step():
    ntt.forwardNTT_batch(a, n, tabs_f2, batch, P, mod); ntt.inverseNTT_batch(a, n, tabs_i2, batch, P, mod)


scratch = a.clone()


def literal_step():
    ntt.forwardNTT_batch(scratch, n, tabs_f, batch, P, lit); ntt.inverseNTT_batch(scratch, n, tabs_i, batch, P, lit)


def rate(fn, k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return batch * k / (e0.elapsed_time(e1) * 1e-3), batch * k / wall


for _ in range(100):
    ctx_step()
for rnd in range(3):
    for name, fn, k in (("context", ctx_step, reps), ("raw checked", checked_step, reps), ("raw trusted", trusted_step, reps),
                        ("raw literal", literal_step, max(3, reps // 10))):
        ev, wall = rate(fn, k)
        print("round %d  %-12s %9.0f pairs/s by events  %9.0f by wall clock" % (rnd, name, ev, wall), flush=True)
assert torch.equal(a, a0)
