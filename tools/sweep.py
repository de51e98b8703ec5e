#!/usr/bin/env python3
"""Throughput sweep over ring degrees and entry points (context API), batch sized to ~256 MiB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ntt_cuda_amd as ntt
import params as P

def find_psi(q, n):
    for x in range(2, 1000):
        psi = pow(x, (q - 1) // (2 * n), q)
        if pow(psi, n, q) == q - 1:
            return psi

dev = torch.device("cuda", 0)
def timeit(f, reps=40):
    # untimed pre-warm flowing straight into the timed launches: a region timed right after a host synchronisation reads
    # the clock ramp (up to 30 % low), not the kernels
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(100): f()
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

for n in (2048, 4096, 8192, 16384, 32768, 65536):
    qs = P.Q60 if n <= 32768 else [P.EDGE_PRIMES[59][0]]
    psis = [find_psi(q, n) for q in qs]
    ctx = ntt.NTTContext(n, qs, psis)
    num = (256 << 20) // (n * 8)
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randint(0, 1 << 58, (num, n), dtype=torch.int64, device=dev, generator=g)
    b = torch.randint(0, 1 << 58, (num, n), dtype=torch.int64, device=dev, generator=g)
    tf = timeit(lambda: ctx.forward_batch(a, num))
    ti = timeit(lambda: ctx.inverse_batch(a, num))
    tm = timeit(lambda: ctx.polymul_batch(a, b, num))
    tp = timeit(lambda: ctx.pointwise_mul(a, a, b, num))
    gb = num * n * 16 / 1e9
    print("n=%6d num=%6d  fwd %.3f ms (%.0f GB/s alg)  inv %.3f ms (%.0f GB/s)  polymul %.3f ms (%.0f GB/s of 3 streams)  pointwise %.3f ms (%.0f GB/s)"
          % (n, num, tf * 1e3, gb / tf, ti * 1e3, gb / ti, tm * 1e3, gb * 1.5 / tm, tp * 1e3, gb * 1.5 / tp))
    ctx.close()
