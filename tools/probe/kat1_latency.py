"""probe: the reference's own decryption_test.cu configuration (n = 4096, its three moduli -- the second one Barrett-inexact: kernel class 0):
us per decryption of its ciphertext, reference words (class 0) and exact kernels.  MI355NTT_LATENCY_PATH_MAX=0 forces the single-pass shape."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ntt-cuda_amd")]
import ntt_cuda_amd as ntt
from ntt_cuda_amd import bfv
z = np.load(os.path.join(ROOT, "tests", "golden", "kat1_decryption_n4096.npz"))
n = int(z["n"]); qs = [int(x) for x in z["q"]]; psis = [int(x) for x in z["psi"]]
for exact in (False, True):
    ctx = bfv.BFVContext(n, qs, psis, int(z["t"]), int(z["gamma"]), exact_on_inexact_primes=exact)
    sk = ntt.to_device(z["sk_host"]); c0 = ntt.to_device(z["c_host"]); c = c0.clone()
    m = ctx.decrypt(c, sk); torch.cuda.synchronize()
    assert np.array_equal(ntt.to_host(m), np.arange(n, dtype=np.uint64) % 10)
    for _ in range(50): ctx.decrypt(c, sk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): ctx.decrypt(c, sk)
    e1.record(); torch.cuda.synchronize()
    print("decryption_test.cu configuration, %s: %.1f us per decryption   (MI355NTT_LATENCY_PATH_MAX=%s)" % (
        "exact kernels (MI355NTT_CTX_EXACT_ON_INEXACT_PRIMES)" if exact else "the reference's words (kernel class 0)", e0.elapsed_time(e1) / 300 * 1e3, os.environ.get("MI355NTT_LATENCY_PATH_MAX")))
    ctx.close()
