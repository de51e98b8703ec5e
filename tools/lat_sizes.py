import os, sys
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, ntt_cuda_amd as ntt, params as P
def find_psi(q, n):
    for x in range(2, 1000):
        psi = pow(x, (q - 1) // (2 * n), q)
        if pow(psi, n, q) == q - 1: return psi
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def lat(f, reps=300):
    for _ in range(50): f()
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for n in (2048, 4096, 8192, 16384, 32768):
    q = P.Q60[0]; psi = find_psi(q, n)
    ctx = ntt.NTTContext(n, [q], [psi])
    for num in (1, 8, 64):
        a = torch.zeros((num, n), dtype=torch.int64, device="cuda:0"); ctx.synth_splitmix(a, num, 1)
        b = a.clone()
        print("n=%6d num=%3d  fwd %6.2f us  inv %6.2f us  polymul %6.2f us" % (n, num, lat(lambda: ctx.forward_batch(a, num)), lat(lambda: ctx.inverse_batch(a, num)), lat(lambda: ctx.polymul_batch(a, b, num))))
    ctx.close()
