#!/usr/bin/env python3
"""Is the n = 2^15 forward+inverse loop power / clock limited?  Sustained runs of (a) one stream x 1024 polynomials, (b) two
streams x 1024 polynomials (the launch ramps of one overlap the tails of the other), (c) four streams x 1024: pairs/s, the
shader clock probed right behind each run, and rocm-smi power / clock samples taken while it runs."""
import os, sys, time, subprocess, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, ROOT)
import torch, ntt_cuda_amd as ntt
from bench import Q60, PSI60
n = 32768
ctx = ntt.NTTContext(n, Q60, PSI60)
dev = torch.device("cuda", 0)

samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), out.strip().replace("\n", " | ")))
        except Exception as e:
            samples.append((time.perf_counter(), "rocm-smi failed: %r" % (e,)))
        time.sleep(0.25)

def run(nstreams, per, seconds=3.0):
    bufs = [torch.empty((per, n), dtype=torch.int64, device=dev) for _ in range(nstreams)]
    for i, b in enumerate(bufs): ctx.synth_splitmix(b, per, 1 + i * per)
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    torch.cuda.synchronize()
    def step():
        for s, b in zip(streams, bufs):
            ctx.forward_batch(b, per, stream=s); ctx.inverse_batch(b, per, stream=s)
    for _ in range(100): step()
    torch.cuda.synchronize()
    reps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50): step()
        reps += 50
        for s in streams: s.synchronize()
    el = time.perf_counter() - t0
    ctx.clock_probe(stream=streams[0]) if hasattr(ctx, "clock_probe") else None
    torch.cuda.synchronize()
    mhz = ctx.probed_clock_mhz() if hasattr(ctx, "probed_clock_mhz") else float("nan")
    return nstreams * per * reps / el, mhz, t0, t0 + el

th = threading.Thread(target=sampler, daemon=True); th.start()
for cfg in [(1, 1024), (2, 1024), (4, 1024), (1, 1024), (2, 512), (1, 256)]:
    r, mhz, ta, tb = run(*cfg)
    mine = [s for (t, s) in samples if ta + 0.5 < t < tb]
    print("streams=%d polys/stream=%4d  => %.3f M pairs/s   probed shader clock %.0f MHz" % (cfg[0], cfg[1], r / 1e6, mhz))
    for s in mine[:3] + mine[-2:]: print("      ", s[-200:])
stop = True
