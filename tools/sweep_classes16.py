#!/usr/bin/env python3
"""n = 65536: throughput of the one-launch forms per kernel class (round 5: classes 5 and 3 have split kernels of their own).
usage: python tools/sweep_classes16.py [num=512]   (GPU; MI355NTT_N16_NO_CLASS35=1 in the environment folds classes 5 / 3 into 4 / 2)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ntt_cuda_amd as ntt
from test_gpu_fuzz_moduli import is_prime, psi_for, STEP

n = 65536
num = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)


def near_primes(k, count):
    out, j = [], 1
    while len(out) < count:
        q = (1 << k) - (STEP * j - 1)
        if is_prime(q):
            out.append(q)
        j += 1
    return out


print("# n = 65536, %d polynomials per launch, context API, warm, back to back%s" % (num, "  [classes 5 / 3 folded into 4 / 2]" if os.environ.get("MI355NTT_N16_NO_CLASS35") else ""))
print("# class                kernel class   fwd ms    inv ms   product ms   TB/s fwd  TB/s inv")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, k in (("hl4-near 60-bit", 60), ("hl5-near 59-bit", 59), ("hl3-near 61-bit", 61), ("hl2-near 62-bit", 62), ("hl6-near 58-bit", 58)):
    qs = near_primes(k, 2)
    psis = [psi_for(q, n) for q in qs]
    ctx = ntt.NTTContext(n, qs, psis)
    a = torch.empty((num, n), dtype=torch.int64, device=dev)
    ctx.synth_splitmix(a, num, 3)
    b = a.clone()
    ref = a.clone()
    ctx.forward_batch(b, num)

    def rate(fn):
        for _ in range(60):
            fn()
        e0.record()
        for _ in range(40):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 40

    f = rate(lambda: ctx.forward_batch(a, num))
    a.copy_(ref)
    ctx.forward_batch(a, num)
    ctx.inverse_batch(a, num)
    torch.cuda.synchronize()
    assert torch.equal(a, ref), name
    i = rate(lambda: ctx.inverse_batch(a, num))
    m = rate(lambda: ctx.polymul_batch(a, b, num))
    by = num * n * 16
    print("  %-18s %-12s %9.4f %9.4f %9.4f %9.2f %9.2f" % (name, ctx.kernel_class, f, i, m, by / f * 1e-9, by / i * 1e-9))
    ctx.close()
