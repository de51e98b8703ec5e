#!/usr/bin/env python3
"""Throughput of the sampler kernels: Salsa20/20 keystream and the byte -> ternary / Gaussian conversion
kernels of keygen_rns / encryption_rns, n = 32768, 4 x 60-bit primes + the special one."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd"))
sys.path.insert(0, ROOT)
import torch
import ntt_cuda_amd as ntt
from ntt_cuda_amd import bfv
from bench import Q60, PSI60, Q60_SPECIAL, PSI60_SPECIAL, BFV_T, BFV_GAMMA

dev = torch.device("cuda", 0)


def rate(fn, reps=50, prewarm=100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(prewarm):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


key = bytes([1] * 32)
for mib in (1, 16, 256):
    out = torch.empty(mib << 20, dtype=torch.uint8, device=dev)
    t = rate(lambda: ntt.salsa20_keystream(out, key, 7), reps=20 if mib > 16 else 50, prewarm=20)
    print("salsa20 keystream %4d MiB: %.3f ms  %.1f GB/s" % (mib, t * 1e3, (mib << 20) / t / 1e9))
n = 32768
qs, psis = Q60 + [Q60_SPECIAL], PSI60 + [PSI60_SPECIAL]
R = len(qs)
ctx = bfv.BFVContext(n, qs, psis, BFV_T, BFV_GAMMA)
rk = torch.empty(ctx.keygen_random_bytes, dtype=torch.uint8, device=dev)
re_ = torch.empty(ctx.encrypt_random_bytes, dtype=torch.uint8, device=dev)
ntt.salsa20_keystream(rk, key, 1)
ntt.salsa20_keystream(re_, key, 2)
sk = torch.empty((R, n), dtype=torch.int64, device=dev)
pk = torch.empty((2, R, n), dtype=torch.int64, device=dev)
tmp = torch.empty((R, n), dtype=torch.int64, device=dev)
c = torch.empty((2, R, n), dtype=torch.int64, device=dev)
e = torch.empty((2, R, n), dtype=torch.int64, device=dev)
t = rate(lambda: ctx.sample_keygen(rk, sk, pk, tmp))
print("sample_keygen  (ternary + uniform + Gaussian, %d random bytes -> %d words): %.2f us" % (rk.numel(), 3 * R * n, t * 1e6))
t = rate(lambda: ctx.sample_encrypt(re_, c, e))
print("sample_encrypt (ternary x2 + Gaussian x2, %d random bytes -> %d words): %.2f us" % (re_.numel(), 4 * R * n, t * 1e6))
t = rate(lambda: ntt.salsa20_keystream(re_, key, 3))
print("keystream for one encryption (%d bytes): %.2f us" % (re_.numel(), t * 1e6))
ctx.close()
