#!/usr/bin/env python3
"""probe (round 5): the n = 2^16 fused product (forward launch + k_inverse15_split<.., MUL>), shipped library against another build (MI355NTT_LIB)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ntt_cuda_amd as ntt, params as P
if os.environ.get("MI355NTT_LIB"):
    ntt.LIB_PATH = os.environ["MI355NTT_LIB"]
dev = torch.device("cuda", 0)
n = 65536
qs = P.Q60
psis = [P.PSI60_65536[q] for q in qs] if hasattr(P, "PSI60_65536") else None
if psis is None:
    import test_gpu_fuzz_moduli as F
    psis = [F.psi_for(q, n) for q in qs]
ctx = ntt.NTTContext(n, qs, psis)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = []
for num, reps in ((256, 300), (512, 200), (2048, 60)):
    a = torch.empty((num, n), dtype=torch.int64, device=dev); ctx.synth_splitmix(a, num, 5)
    b = a.clone(); ctx.forward_batch(b, num)
    def rate(fn):
        for _ in range(reps): fn()
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    out.append("%5d: fused %.4f ms  inverse %.4f ms" % (num, rate(lambda: ctx.polymul_batch(a, b, num)), rate(lambda: ctx.inverse_batch(a, num))))
print("# n = 65536  lib = %s :  %s" % (os.path.basename(os.environ.get("MI355NTT_LIB") or "shipped"), "   ".join(out)))
