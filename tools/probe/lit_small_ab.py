"""probe: class-0 contexts (Barrett-inexact 60-bit modulus), batch sizes 1 ... 512: us per forward + inverse pair and per fused product.
Run once per path: MI355NTT_LATENCY_PATH_MAX=0 (single-pass literal kernels at every size) / =1000000 (small-batch literal kernels at
every size) / unset (the library's switching points)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ntt-cuda_amd"), os.path.join(ROOT, "tests")]
import ntt_cuda_amd as ntt
import params as P

def timeit(fn, reps=int(os.environ.get('LIT_AB_REPS', 100)), warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

tag = os.environ.get("MI355NTT_LATENCY_PATH_MAX", "default")
for n in (2048, 4096, 8192, 16384, 32768):
    q, roots = P.INEXACT_PRIMES[60]
    ctx = ntt.NTTContext(n, [q], [roots[n]])
    default = (1, 4, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512)
    fine = {2048: (512, 768, 1024, 1536, 2048, 3072, 4096, 8192), 4096: (512, 768, 1024, 1536, 2048, 4096), 8192: (384, 512, 640, 768, 1024, 2048),
            16384: (256, 320, 384, 448, 512, 640, 768), 32768: (192, 224, 256, 288, 320, 384, 448, 480, 512, 640)}
    big = {2048: (8192, 16384, 32768), 4096: (4096, 8192, 16384), 8192: (2048, 4096, 8192), 16384: (1024, 2048, 4096), 32768: (768, 1024, 1280, 2048, 4096, 8192)}
    for num in (big[n] if os.environ.get("LIT_AB_BIG") else fine[n] if os.environ.get("LIT_AB_FINE") else default):
        a = torch.randint(0, q, (num, n), dtype=torch.int64, device="cuda")
        b = torch.randint(0, q, (num, n), dtype=torch.int64, device="cuda")
        def pair():
            ctx.forward_batch(a, num); ctx.inverse_batch(a, num)
        def mul():
            ctx.polymul_batch(a, b, num)
        print("path %-8s n %6d batch %4d  pair %8.1f us  product %8.1f us" % (tag, n, num, timeit(pair), timeit(mul)), flush=True)
    ctx.close()
