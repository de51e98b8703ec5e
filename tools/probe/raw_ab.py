"""checked vs trusted raw calls on the bench workload (n = 32768, 4 x 60-bit primes, 1024 polynomials): pairs/s, three chunks each"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ntt-cuda_amd")]
import ntt_cuda_amd as ntt
sys.path.insert(0, ROOT)
import bench
n, P, batch = 32768, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
ctx = ntt.NTTContext(n, bench.Q60, bench.PSI60)
a = bench.synth_recipe(torch, ctx, batch, n, dev, seed_base=1)
a0 = a.clone()
tabs_f = torch.empty((P, n), dtype=torch.int64, device=dev); tabs_i = torch.empty((P, n), dtype=torch.int64, device=dev)
for i in range(P):
    tp, ti = ntt.fillTablePsi128(bench.PSI60[i], bench.Q60[i], ntt.modinv128(bench.PSI60[i], bench.Q60[i]), n)
    tabs_f[i] = torch.from_numpy(tp.view(np.int64)); tabs_i[i] = torch.from_numpy(ti.view(np.int64))
mod = ntt.Moduli(bench.Q60)
def raw_step():
    ntt.forwardNTT_batch(a, n, tabs_f, batch, P, mod); ntt.inverseNTT_batch(a, n, tabs_i, batch, P, mod)
def ctx_step():
    ctx.forward_batch(a, batch); ctx.inverse_batch(a, batch)
def rate(fn, reps=60, prewarm=600):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for _ in range(prewarm): fn()
    ev[0].record()
    for c in range(3):
        for _ in range(reps): fn()
        ev[c + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[c].elapsed_time(ev[c + 1]) for c in range(3))
    return batch * reps / (ms[1] * 1e-3)
r_ctx = rate(ctx_step)
r_chk = rate(raw_step)
assert torch.equal(a, a0)
r_ctx2 = rate(ctx_step)
r_chk2 = rate(raw_step)
print("context %.4f M  checked raw %.4f M  (%.3f)   again: %.4f M / %.4f M (%.3f)   MI355NTT_NO_IN_KERNEL_CHECK=%s" % (
    r_ctx / 1e6, r_chk / 1e6, r_chk / r_ctx, r_ctx2 / 1e6, r_chk2 / 1e6, r_chk2 / r_ctx2, os.environ.get("MI355NTT_NO_IN_KERNEL_CHECK")))
