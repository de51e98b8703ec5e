#!/usr/bin/env python3
"""probe: n = 65536 forward, pair launch against the single-workgroup form (MI355NTT_NO_PAIR16=1), one and two primes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ntt_cuda_amd as ntt, params as P
if os.environ.get('MI355NTT_LIB'):
    ntt.LIB_PATH = os.environ['MI355NTT_LIB']
dev = torch.device("cuda", 0)
n, num = 65536, 512
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for qs in (P.Q60[:1],):
    psis = [next(x for x in (pow(g, (q - 1) // (2 * n), q) for g in range(2, 2000)) if pow(x, n, q) == q - 1) for q in qs]
    ctx = ntt.NTTContext(n, qs, psis)
    a = torch.empty((num, n), dtype=torch.int64, device=dev); ctx.synth_splitmix(a, num, 5)
    for rep in range(3):
        for _ in range(60): ctx.forward_batch(a, num)
        e0.record()
        for _ in range(40): ctx.forward_batch(a, num)
        e1.record(); torch.cuda.synchronize()
        print("lib %s primes %d  NO_PAIR16=%s  forward %.4f ms" % (os.path.basename(ntt.LIB_PATH), len(qs), os.environ.get("MI355NTT_NO_PAIR16"), e0.elapsed_time(e1) / 40))
    ctx.close()
