#!/usr/bin/env python3
"""n = 2^16: forward / inverse / product time against the batch size (run twice: default dispatch, and with
MI355NTT_LATENCY_PATH_MAX=0 in the environment = fused / pair launches at every size) -> where the dispatch should switch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ntt_cuda_amd as ntt
import params as P

n = 65536
q = P.Q60[0]
psi = next(pw for pw in (pow(x, (q - 1) // (2 * n), q) for x in range(2, 1000)) if pow(pw, n, q) == q - 1)
dev = torch.device("cuda", 0)
ctx = ntt.NTTContext(n, [q], [psi])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

def timeit(f, reps=100):
    for _ in range(100): f()
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

print("mode:", "MI355NTT_LATENCY_PATH_MAX=" + os.environ.get("MI355NTT_LATENCY_PATH_MAX", "(default)"))
for num in (1, 8, 16, 32, 48, 64, 80, 88, 96, 112, 128, 144, 160, 176, 192, 256, 384, 512):
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randint(0, 1 << 58, (num, n), dtype=torch.int64, device=dev, generator=g)
    b = torch.randint(0, 1 << 58, (num, n), dtype=torch.int64, device=dev, generator=g)
    tf = timeit(lambda: ctx.forward_batch(a, num)); ti = timeit(lambda: ctx.inverse_batch(a, num)); tm = timeit(lambda: ctx.polymul_batch(a, b, num))
    print("num=%4d  fwd %7.1f us  inv %7.1f us  product %7.1f us" % (num, tf, ti, tm))
