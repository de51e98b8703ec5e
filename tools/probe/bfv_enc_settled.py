"""batched BFV encryption and decryption behind settled clocks (30 untimed calls, then 30 timed): us per call for 64 ... 512 ciphertexts"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ntt-cuda_amd"), ROOT]
import bench, ntt_cuda_amd as ntt
from ntt_cuda_amd import bfv
dev = torch.device("cuda", 0)
n = 32768
qs, psis = bench.Q60 + [bench.Q60_SPECIAL], bench.PSI60 + [bench.PSI60_SPECIAL]
R = len(qs)
ctx = bfv.BFVContext(n, qs, psis, bench.BFV_T, bench.BFV_GAMMA)
pk = bench.synth(torch, 2 * R, n, qs, dev, seed=5).reshape(2, R, n)
for count in (64, 128, 256):
    c = bench.synth(torch, 2 * count * R, n, qs, dev, seed=3).reshape(2, count, R, n)
    e = bench.synth(torch, 2 * count * R, n, qs, dev, seed=6).reshape(2, count, R, n)
    m = torch.randint(0, bench.BFV_T, (count, n), dtype=torch.int64, device=dev)
    sk = bench.synth(torch, R, n, qs, dev, seed=4)
    def t(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(30): fn()
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 30 * 1e3
    enc = t(lambda: ctx.encrypt_batch(c, pk, e, m, count))
    dec = t(lambda: ctx.decrypt_batch(c, sk, count))
    print("%4d ciphertexts per call: encrypt %.1f us (%.2f us each)   decrypt %.1f us (%.2f us each)" % (count, enc, enc / count, dec, dec / count), flush=True)
