#!/usr/bin/env python3
"""n = 2^15, four 60-bit primes: forward / inverse / forward+inverse pairs / fused products over batch sizes from the
Infinity-Cache-resident bench batch (1024 polynomials = 256 MiB) to configs[3]'s global batch resident on one GPU
(8192 = 2 GiB).  HIP events around back-to-back launches behind an untimed pre-warm.  -> profiles/r04_sweep_batches.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ntt_cuda_amd as ntt
import params as P

dev = torch.device("cuda", 0)
n = 32768
qs = P.Q60
psis = [P.find_psi(q, n) if hasattr(P, "find_psi") else None for q in qs]
if psis[0] is None:
    def find_psi(q, n):
        for x in range(2, 1000):
            psi = pow(x, (q - 1) // (2 * n), q)
            if pow(psi, n, q) == q - 1:
                return psi
    psis = [find_psi(q, n) for q in qs]
ctx = ntt.NTTContext(n, qs, psis)

def timeit(f, reps, warm):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm): f()
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

batches = [int(x) for x in sys.argv[1:]] or [256, 512, 1024, 2048, 4096, 8192]
print("n = 32768, 4 x 60-bit primes; ms per launch (ms per 1024 polynomials) [TB/s algorithmic]")
for num in batches:
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randint(0, 1 << 58, (num, n), dtype=torch.int64, device=dev, generator=g)
    b = torch.randint(0, 1 << 58, (num, n), dtype=torch.int64, device=dev, generator=g)
    reps = max(10, 40960 // num); warm = max(10, 102400 // num)
    for rnd in range(2):
        tf = timeit(lambda: ctx.forward_batch(a, num), reps, warm)
        ti = timeit(lambda: ctx.inverse_batch(a, num), reps, warm)
        def pair():
            ctx.forward_batch(a, num); ctx.inverse_batch(a, num)
        tp = timeit(pair, reps, warm)
        tm = timeit(lambda: ctx.polymul_batch(a, b, num), reps, warm)
        by = num * n * 16
        print("batch %5d round %d: fwd %.4f (%.4f) [%.2f]  inv %.4f (%.4f) [%.2f]  pair %.4f (%.4f) = %.3f M pairs/s  polymul %.4f (%.4f) = %.3f M/s, %.3f x pair"
              % (num, rnd, tf * 1e3, tf * 1e3 * 1024 / num, by / tf / 1e12, ti * 1e3, ti * 1e3 * 1024 / num, by / ti / 1e12,
                 tp * 1e3, tp * 1e3 * 1024 / num, num / tp / 1e6, tm * 1e3, tm * 1e3 * 1024 / num, num / tm / 1e6, tm / tp), flush=True)
    del a, b
ctx.close()
