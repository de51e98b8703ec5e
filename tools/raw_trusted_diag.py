#!/usr/bin/env python3
"""Why did the first timed region of trusted raw calls read low?  Per-chunk timings (10 steps per chunk, HIP
events, no host synchronisation in between) across: context API -> checked raw calls -> mi355ntt_raw_trust_tables -> trusted raw calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import ntt_cuda_amd as ntt
from bench import Q60, PSI60, synth
n, P, batch = 32768, 4, 1024
dev = torch.device("cuda", 0)
ctx = ntt.NTTContext(n, Q60, PSI60)
a = synth(torch, batch, n, Q60, dev, 1)
tabs_f = torch.empty((P, n), dtype=torch.int64, device=dev); tabs_i = torch.empty((P, n), dtype=torch.int64, device=dev)
for i in range(P):
    tp, ti = ntt.fillTablePsi128(PSI60[i], Q60[i], ntt.modinv128(PSI60[i], Q60[i]), n)
    tabs_f[i] = torch.from_numpy(tp.view(np.int64)); tabs_i[i] = torch.from_numpy(ti.view(np.int64))
mod = ntt.Moduli(Q60)
def ctx_step(): ctx.forward_batch(a, batch); ctx.inverse_batch(a, batch)
def raw_step(): ntt.forwardNTT_batch(a, n, tabs_f, batch, P, mod); ntt.inverseNTT_batch(a, n, tabs_i, batch, P, mod)
def chunks(fn, k, label, sync_first):
    if sync_first: torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for c in range(k):
        for _ in range(10): fn()
        ev[c + 1].record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    ms = [ev[c].elapsed_time(ev[c + 1]) / 10 for c in range(k)]
    print("%-46s host enqueue %.1f ms for %d steps | ms/step per chunk: %s" % (label, t_host * 1e3, 10 * k, " ".join("%.3f" % m for m in ms)), flush=True)
for _ in range(300): ctx_step()
chunks(ctx_step, 8, "context API (hot)", False)
chunks(raw_step, 12, "raw checked, first sight (derives the context)", False)
chunks(raw_step, 12, "raw checked again", False)
chunks(raw_step, 12, "raw checked after a synchronise", True)
ntt.raw_trust_tables(n, tabs_f, mod); ntt.raw_trust_tables(n, tabs_i, mod, inverse=True)
chunks(raw_step, 16, "raw trusted right after the trust calls", False)
chunks(raw_step, 12, "raw trusted after a synchronise", True)
chunks(ctx_step, 12, "context API after a synchronise", True)
tabs_f2, tabs_i2 = tabs_f.clone(), tabs_i.clone()
def raw2_step(): ntt.forwardNTT_batch(a, n, tabs_f2, batch, P, mod); ntt.inverseNTT_batch(a, n, tabs_i2, batch, P, mod)
t0 = time.perf_counter(); ok = ntt.raw_trust_tables(n, tabs_f2, mod) and ntt.raw_trust_tables(n, tabs_i2, mod, inverse=True); print("trust of a second table pair: %s, %.1f ms on the host" % (ok, (time.perf_counter() - t0) * 1e3))
chunks(raw2_step, 16, "raw trusted, second table pair, first calls", False)
chunks(raw2_step, 12, "raw trusted, second table pair, again", False)
