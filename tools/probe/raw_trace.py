"""a short train of checked raw calls for rocprofv3 --kernel-trace (the comparison kernel in front: MI355NTT_NO_IN_KERNEL_CHECK=1)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ntt-cuda_amd"), ROOT]
import ntt_cuda_amd as ntt
import bench
n, P, batch = 32768, 4, 1024
dev = torch.device("cuda", 0)
ctx = ntt.NTTContext(n, bench.Q60, bench.PSI60)
a = bench.synth_recipe(torch, ctx, batch, n, dev, seed_base=1)
tabs_f = torch.empty((P, n), dtype=torch.int64, device=dev); tabs_i = torch.empty((P, n), dtype=torch.int64, device=dev)
for i in range(P):
    tp, ti = ntt.fillTablePsi128(bench.PSI60[i], bench.Q60[i], ntt.modinv128(bench.PSI60[i], bench.Q60[i]), n)
    tabs_f[i] = torch.from_numpy(tp.view(np.int64)); tabs_i[i] = torch.from_numpy(ti.view(np.int64))
mod = ntt.Moduli(bench.Q60)
for _ in range(400):
    ntt.forwardNTT_batch(a, n, tabs_f, batch, P, mod); ntt.inverseNTT_batch(a, n, tabs_i, batch, P, mod)
torch.cuda.synchronize()
