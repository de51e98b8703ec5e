"""batched BFV decryption with and without the scaling in the fused product's store path (MI355NTT_NO_FUSED_EPILOGUE=1): us per call"""
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
out = []
for count in (64, 128, 256, 512):
    c = bench.synth(torch, 2 * count * R, n, qs, dev, seed=3).reshape(2, count, R, n)
    sk = bench.synth(torch, R, n, qs, dev, seed=4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(30):
        ctx.decrypt_batch(c, sk, count)
    e0.record()
    for _ in range(30):
        ctx.decrypt_batch(c, sk, count)
    e1.record()
    torch.cuda.synchronize()
    out.append("%d: %.1f us" % (count, e0.elapsed_time(e1) / 30 * 1e3))
print("decrypt_batch per call  ", "  ".join(out), "  NO_FUSED_EPILOGUE=%s" % os.environ.get("MI355NTT_NO_FUSED_EPILOGUE"))
