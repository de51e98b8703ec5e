"""GPU: bit-exact parity of the HIP path (called through the C ABI) against the CPU oracle, the committed
golden vectors and size-independent properties at BASELINE.json's full sizes.  Integer work: the bar is
equality of every word."""
import ctypes
import os

import numpy as np
import pytest

import params as P

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat1_decryption_n4096.npz")


def dev(native, a):
    return native.to_device(a)


def host(native, t):
    import torch
    torch.cuda.synchronize()
    return native.to_host(t)


# ------------------------------------------------------------------------------------------ raw API
@pytest.mark.parametrize("n", sorted(P.REF_PARAMS))
def test_raw_forward_inverse_reference_params(native, oracle, gpu, n):
    """forwardNTT / inverseNTT (ntt_60bit.cuh:314,350) with the reference's own getParams sets."""
    import torch
    q, psi, psiinv, ninv, qbit = native.getParams(n)
    prm = oracle.Params(n, [q], [psi])
    assert int(prm.k[0]) == qbit and int(prm.psiinv[0]) == psiinv
    mu = native.barrett_mu(q, qbit)
    tp, ti = native.fillTablePsi128(psi, q, psiinv, n)
    d_tp, d_ti = dev(native, tp), dev(native, ti)
    a = oracle.splitmix(n, 1, q)
    d_a = dev(native, a)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        native.forwardNTT(d_a, n, s, q, mu, qbit, d_tp)
    s.synchronize()
    A = oracle.forward(a, prm)
    assert np.array_equal(host(native, d_a), A)
    native.inverseNTT(d_a, n, None, q, mu, qbit, d_ti)
    assert np.array_equal(host(native, d_a), a)


def test_raw_polymul_like_60bit_ntt_test(native, oracle, gpu):
    """60bit_ntt_test.cu:72-98 with check = 1: forwardNTTdouble, barrett, inverseNTT vs refPolyMul128, N = 2048."""
    import torch
    n = 2048
    q, psi, psiinv, ninv, qbit = native.getParams(n)
    mu = native.barrett_mu(q, qbit)
    tp, ti = native.fillTablePsi128(psi, q, psiinv, n)
    d_tp, d_ti = dev(native, tp), dev(native, ti)
    a, b = oracle.splitmix(n, 21, q), oracle.splitmix(n, 22, q)
    refc = oracle.ref_polymul(a, b, q)
    d_a, d_b = dev(native, a), dev(native, b)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    native.forwardNTTdouble(d_a, d_b, n, s1, s2, q, mu, qbit, d_tp)
    torch.cuda.synchronize()
    native.barrett(d_a, d_b, q, mu, qbit)
    native.inverseNTT(d_a, n, None, q, mu, qbit, d_ti)
    assert np.array_equal(host(native, d_a), refc)
    # half_poly_mul_device / full_poly_mul compositions (poly_arithmetic.cuh:277-310)
    d_a2 = dev(native, a)
    native.half_poly_mul_device(d_a2, d_b, n, None, q, mu, qbit, d_tp, d_ti)   # d_b is already NTT(b)
    assert np.array_equal(host(native, d_a2), refc)
    assert np.array_equal(native.full_poly_mul(a, b, n, q, mu, qbit, d_tp, d_ti), refc)


def test_raw_batch_kat1_decryption_vectors(native, oracle, gpu):
    """The reference's only pinned vectors (decryption_test.cu:348,355): forwardNTT_batch -> barrett_batch ->
    inverseNTT_batch on c1 with num = r, division = r + 1 (bfv_decryption.cuh:98-101) reproduces the golden
    intermediate digests; the rest of decryption_rns (oracle) then yields m[i] = i % 10."""
    z = np.load(GOLD)
    n, r = int(z["n"]), 2
    q, psi = z["q"], z["psi"]
    prm = oracle.Params(n, q, psi)                     # all r+1 primes: tables are [r+1][n]
    mod = native.Moduli(q)
    c = z["c_host"].copy()
    d_c = dev(native, c)
    d_sk = dev(native, z["sk_host"])
    d_tp, d_ti = dev(native, prm.psi_tabs), dev(native, prm.psiinv_tabs)
    c1 = d_c[(r + 1) * n:]
    native.forwardNTT_batch(c1, n, d_tp, r, r + 1, mod)
    assert P.digest(host(native, c1)[: r * n]) == P.KAT1_STAGE_DIGESTS[0]
    native.barrett_batch(c1, d_sk, n, r, mod, num=r)
    assert P.digest(host(native, c1)[: r * n]) == P.KAT1_STAGE_DIGESTS[1]
    native.inverseNTT_batch(c1, n, d_ti, r, r + 1, mod)
    got = host(native, c1)[: r * n]
    assert P.digest(got) == P.KAT1_STAGE_DIGESTS[2]
    _, stages = oracle.bfv_decrypt(z["c_host"], z["sk_host"], q, psi, n, int(z["t"]), int(z["gamma"]), want_stages=True)
    assert np.array_equal(got, stages[2])


def test_raw_barrett_variants(native, oracle, gpu):
    n = 4096
    qs = [P.REF_PARAMS_4096_58BIT[0], P.REF_PARAMS[4096][0], P.EDGE_PRIMES[61][0]]
    psis = [P.REF_PARAMS_4096_58BIT[1], P.REF_PARAMS[4096][1], P.EDGE_PRIMES[61][1][4096]]
    prm = oracle.Params(n, qs, psis, tables=False)
    mod = native.Moduli(qs)
    a, b = oracle.synth_batch(n, 5, qs, 100), oracle.synth_batch(n, 5, qs, 200)
    want = oracle.pointwise_batch(a, b, prm, division=3)
    d_a, d_b = dev(native, a), dev(native, b)
    d_c = dev(native, np.zeros_like(a))
    native.barrett_batch_3param(d_c, d_a, d_b, n, 3, mod)
    assert np.array_equal(host(native, d_c), want) and np.array_equal(host(native, d_a), a)
    native.barrett_batch(d_a, d_b, n, 3, mod)
    assert np.array_equal(host(native, d_a), want)
    x = oracle.splitmix(n, 9, qs[2])
    d_x = dev(native, x)
    native.barrett_int(d_x, qs[2] - 5, qs[2], int(prm.mu[2]), int(prm.k[2]))
    assert np.array_equal(host(native, d_x), oracle.pointwise_scalar(x, qs[2] - 5, prm, 2))


# -------------------------------------------------------------------------------------- context API
CTX_CASES = [
    (4096, [P.REF_PARAMS_4096_58BIT[0]], [P.REF_PARAMS_4096_58BIT[1]]),       # BASELINE config 1 (58-bit)
    (4096, [P.REF_PARAMS[4096][0]], [P.REF_PARAMS[4096][1]]),                  # the reference's active 25-bit set
    (32768, [P.REF_PARAMS[32768][0]], [P.REF_PARAMS[32768][1]]),               # reference 55-bit
    (32768, P.Q60[:1], P.PSI60[:1]),                                            # BASELINE config 2
    (32768, P.Q60, P.PSI60),                                                    # BASELINE config 3/4 primes
    (32768, P.Q55, P.PSI55),                                                    # demo.cu:35-36
    (32768, [P.EDGE_PRIMES[b][0] for b in (62, 61, 59, 30)], [P.EDGE_PRIMES[b][1][32768] for b in (62, 61, 59, 30)]),
    (65536, [P.EDGE_PRIMES[61][0]], [P.EDGE_PRIMES[61][1][65536]]),
    (65536, [P.EDGE_PRIMES[b][0] for b in (62, 59, 30)], [P.EDGE_PRIMES[b][1][65536] for b in (62, 59, 30)]),   # n = 2^16 runs as 2 x 2^15
    (2048, [P.REF_PARAMS[2048][0]], [P.REF_PARAMS[2048][1]]),
    (8192, [P.REF_PARAMS[8192][0]], [P.REF_PARAMS[8192][1]]),
    (16384, [P.REF_PARAMS[16384][0]], [P.REF_PARAMS[16384][1]]),
]


@pytest.mark.parametrize("n,qs,psis", CTX_CASES, ids=lambda v: str(v) if isinstance(v, int) else None)
def test_ctx_batch_matches_oracle(native, oracle, gpu, n, qs, psis):
    """forward_batch / pointwise / inverse_batch / fused polymul vs the oracle, ragged batch (num % P != 0)."""
    Pn = len(qs)
    num = 2 * Pn + 1
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    for i in range(Pn):
        info = ctx.prime(i)
        assert (info["q"], info["mu"], info["bit_length"], info["psiinv"]) == (qs[i], int(prm.mu[i]), int(prm.k[i]), int(prm.psiinv[i]))
    a, b = oracle.synth_batch(n, num, qs, 1), oracle.synth_batch(n, num, qs, 1000)
    A, B = oracle.forward_batch(a, prm), oracle.forward_batch(b, prm)
    AB = oracle.pointwise_batch(A, B, prm)
    C = oracle.inverse_batch(AB, prm)
    d_a, d_b = dev(native, a), dev(native, b)
    ctx.forward_batch(d_a, num)
    assert np.array_equal(host(native, d_a), A)
    ctx.forward_batch(d_b, num)
    d_c = dev(native, np.zeros_like(a))
    ctx.pointwise_mul(d_c, d_a, d_b, num)
    assert np.array_equal(host(native, d_c), AB)
    ctx.inverse_batch(d_c, num)
    assert np.array_equal(host(native, d_c), C)
    ctx.inverse_batch(d_a, num)
    assert np.array_equal(host(native, d_a), a)
    # fused NTT -> (.) -> INTT equals the three-call composition (bfv_encryption.cuh:268-271)
    d_a2 = dev(native, a)
    ctx.polymul_batch(d_a2, d_b, num)
    assert np.array_equal(host(native, d_a2), C)
    # single-polynomial entry points, every prime
    for i in range(Pn):
        x = oracle.splitmix(n, 77 + i, qs[i])
        d_x = dev(native, x)
        ctx.forward(d_x, i)
        assert np.array_equal(host(native, d_x), oracle.forward(x, prm, i))
        ctx.inverse(d_x, i)
        assert np.array_equal(host(native, d_x), x)
    ctx.close()


@pytest.mark.parametrize("key", sorted(P.GOLDEN_DIGESTS))
def test_ctx_golden_digests(native, oracle, gpu, key):
    """SURVEY.md 4.2 digests reproduced on the GPU (table upload, NTT(a), negacyclic product)."""
    n, q, psi = key
    d_tab, d_ntt, d_mul = P.GOLDEN_DIGESTS[key]
    ctx = native.NTTContext(n, [q], [psi])
    import torch
    tab = torch.empty(n, dtype=torch.int64, device=gpu)
    ctypes.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(ctypes.c_void_p(tab.data_ptr()), ctypes.c_void_p(ctx.psi_tables_ptr),
                                                        ctypes.c_size_t(n * 8), 3)
    assert P.digest(host(native, tab)) == d_tab
    a, b = oracle.splitmix(n, 1, q), oracle.splitmix(n, 2, q)
    d_a, d_b = dev(native, a), dev(native, b)
    ctx.forward_double(d_a, d_b)
    assert P.digest(host(native, d_a)) == d_ntt
    ctx.pointwise_mul(d_a, d_a, d_b, 1)
    ctx.inverse(d_a)
    assert P.digest(host(native, d_a)) == d_mul
    ctx.close()


def test_ctx_edge_inputs(native, oracle, gpu):
    """zeros, all q-1, delta polynomials, empty batch, division < num_primes, scalar multiply."""
    n, qs, psis = 32768, P.Q60, P.PSI60
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    num = 8
    a = np.zeros((num, n), dtype=np.uint64)
    for y in range(num):
        q = qs[y % 4]
        if y < 4:
            a[y, :] = q - 1                      # maximum residues everywhere
        elif y == 4:
            a[y, 0] = 1                          # delta: NTT is all ones
        elif y == 5:
            a[y, n - 1] = q - 1
        elif y == 6:
            a[y, ::2] = q - 1
    d_a = dev(native, a)
    ctx.forward_batch(d_a, num)
    A = host(native, d_a).reshape(num, n)
    assert np.array_equal(A, oracle.forward_batch(a, prm).reshape(num, n))
    assert np.all(A[4] == 1) and np.all(A[7] == 0)
    ctx.inverse_batch(d_a, num)
    assert np.array_equal(host(native, d_a).reshape(num, n), a)
    # empty batch is a no-op, not an error
    ctx.forward_batch(d_a, 0)
    ctx.inverse_batch(d_a, 0)
    ctx.polymul_batch(d_a, d_a, 0)
    assert np.array_equal(host(native, d_a).reshape(num, n), a)
    # division = 2 uses only primes 0,1
    sub = oracle.Params(n, qs[:2], psis[:2])
    x = oracle.synth_batch(n, 5, qs[:2], 500)
    d_x = dev(native, x)
    ctx.forward_batch(d_x, 5, division=2)
    assert np.array_equal(host(native, d_x), oracle.forward_batch(x, sub))
    with pytest.raises(native.NTTError):
        ctx.forward_batch(d_x, 5, division=5)
    # barrett_int
    y = oracle.splitmix(n, 5, qs[3])
    d_y = dev(native, y)
    ctx.pointwise_mul_scalar(d_y, qs[3] - 1, 3)
    assert np.array_equal(host(native, d_y), oracle.pointwise_scalar(y, qs[3] - 1, prm, 3))
    ctx.close()


# ---------------------------------------------------------------- full BASELINE sizes: properties
@pytest.mark.parametrize("num", [256, 1024])
def test_full_size_properties(native, oracle, gpu, num):
    """BASELINE configs 3 (batch 256) and 4's per-GPU shard (batch 1024), n = 32768, 4 x 60-bit RNS:
    round trip, linearity, fused == unfused, and oracle equality on a sample of polynomials."""
    import torch
    n, qs, psis = 32768, P.Q60, P.PSI60
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    qvec = torch.tensor(np.array(qs, dtype=np.uint64).view(np.int64), device=gpu)
    qcol = qvec[torch.arange(num, device=gpu) % 4].unsqueeze(1)          # modulus of each polynomial
    # device-side synthetic residues: 60-bit randoms reduced once (values < 2^60 < 2q -> one subtraction)
    g = torch.Generator(device=gpu).manual_seed(1234 + num)
    a = torch.randint(0, 1 << 60, (num, n), dtype=torch.int64, device=gpu, generator=g)
    b = torch.randint(0, 1 << 60, (num, n), dtype=torch.int64, device=gpu, generator=g)
    a = torch.where(a >= qcol, a - qcol, a)
    b = torch.where(b >= qcol, b - qcol, b)
    a0, b0 = a.clone(), b.clone()
    # round trip
    ctx.forward_batch(a, num)
    A = a.clone()
    ctx.inverse_batch(a, num)
    assert torch.equal(a, a0)
    # oracle equality on a sample
    sample = [0, 1, 2, 3, num // 2 + 1, num - 2, num - 1]
    hostA = native.to_host(A[sample].contiguous())
    for row, y in enumerate(sample):
        assert np.array_equal(hostA[row], oracle.forward(native.to_host(a0[y].contiguous()), prm, y % 4)), y
    # linearity: NTT(a + b) = NTT(a) + NTT(b)  (mod q)
    s = a0 + b0
    s = torch.where(s >= qcol, s - qcol, s)
    ctx.forward_batch(s, num)
    ctx.forward_batch(b, num)
    t = A + b
    t = torch.where(t >= qcol, t - qcol, t)
    assert torch.equal(s, t)
    # outputs are canonical
    assert bool((A < qcol).all()) and bool((A >= 0).all())
    # fused polymul == forward, pointwise, inverse
    c = A.clone()
    ctx.pointwise_mul(c, A, b, num)
    ctx.inverse_batch(c, num)
    f = a0.clone()
    ctx.polymul_batch(f, b, num)
    assert torch.equal(f, c)
    # checksum of checksums against the oracle for the product on the sample
    hostC = native.to_host(c[sample].contiguous())
    for row, y in enumerate(sample):
        want = oracle.inverse(oracle.pointwise_batch(oracle.forward(native.to_host(a0[y].contiguous()), prm, y % 4),
                                                     oracle.forward(native.to_host(b0[y].contiguous()), prm, y % 4),
                                                     oracle.Params(n, [qs[y % 4]], [psis[y % 4]], tables=False)), prm, y % 4)
        assert np.array_equal(hostC[row], want), y
    ctx.close()


@pytest.mark.parametrize("num", [257, 777])
def test_persistent_loop_ragged_counts(native, oracle, gpu, num):
    """n = 2^15 kernels are persistent (grid = min(num, #CUs), each workgroup walks y, y + grid, ...): batch sizes that
    are not multiples of the grid or of the prime count must still transform every polynomial exactly once."""
    import torch
    n, qs, psis = 32768, P.Q60, P.PSI60
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    a = native.to_device(oracle.synth_batch(n, num, qs, 4242))
    b = native.to_device(oracle.synth_batch(n, num, qs, 9000))
    a0 = a.clone()
    ctx.forward_batch(a, num)
    A = a.clone()
    sample = [0, 1, 255, 256, num - 2, num - 1]
    for y in sample:
        assert np.array_equal(native.to_host(A[y].contiguous()), oracle.forward(native.to_host(a0[y].contiguous()), prm, y % 4)), y
    ctx.inverse_batch(a, num)
    assert torch.equal(a, a0)
    ctx.forward_batch(b, num)
    c = A.clone()
    ctx.pointwise_mul(c, A, b, num)
    ctx.inverse_batch(c, num)
    f = a0.clone()
    ctx.polymul_batch(f, b, num)
    assert torch.equal(f, c)
    ctx.close()


KERNEL_FORMS = {
    "hl6-near": (P.Q55, P.PSI55),
    "hl6-general": ([P.GENERAL_PRIMES[57][0], P.Q55[0]], [P.GENERAL_PRIMES[57][1], P.PSI55[0]]),
    "hl4-near": (P.Q60, P.PSI60),
    "hl4-general": ([P.GENERAL_PRIMES[60][0], P.GENERAL_PRIMES[59][0], P.Q60[0]], [P.GENERAL_PRIMES[60][1], P.GENERAL_PRIMES[59][1], P.PSI60[0]]),
    "hl2-near": ([P.EDGE_PRIMES[62][0], P.EDGE_PRIMES[61][0]], [P.EDGE_PRIMES[62][1][32768], P.EDGE_PRIMES[61][1][32768]]),
    "hl2-general": ([P.GENERAL_PRIMES[62][0], P.Q60[1]], [P.GENERAL_PRIMES[62][1], P.PSI60[1]]),
    # round 4: classes of their own for 61-bit and 59-bit near-2^k moduli (dispatch_class, kernels_fast_impl.cuh)
    "hl3-near": ([P.EDGE_PRIMES[61][0]], [P.EDGE_PRIMES[61][1][32768]]),
    "hl5-near": ([P.EDGE_PRIMES[59][0], P.EDGE_PRIMES[59][0]], [P.EDGE_PRIMES[59][1][32768], P.EDGE_PRIMES[59][1][32768]]),
    # round 5: general 61-bit primes leave class 2 (three-product quotients, one conditional subtraction of 4q per reducing sum)
    "hl3-general": (sorted(P.GENERAL61), [P.GENERAL61[q][32768] for q in sorted(P.GENERAL61)]),
}


@pytest.mark.parametrize("form", sorted(KERNEL_FORMS))
@pytest.mark.parametrize("num", [7, 200, 300])
def test_every_kernel_form_at_n32768_matches_oracle(native, oracle, gpu, form, num):
    """n = 2^15 has eight instantiations per kernel (headroom class x near-2^k or general prime: partial reduction, exact or
    approximate quotient, fused butterfly) on two paths (the small-batch kernels of kernels_lat.cuh for 7 and -- just above one
    polynomial per CU -- 300 polynomials, the persistent single-pass kernels for 200; use_latency_path, kernels_fast_impl.cuh):
    every one of them against the oracle, with adversarial coefficients (0, 1, q-1, q-2) mixed into the random ones."""
    import torch
    n = 32768
    qs, psis = KERNEL_FORMS[form]
    Pn = len(qs)
    for q in qs:
        assert native.barrett_is_exact(q)
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    assert not ctx.uses_literal_kernels
    a = oracle.synth_batch(n, num, qs, 31337).reshape(num, n)
    b = oracle.synth_batch(n, num, qs, 555).reshape(num, n)
    for y in range(min(num, 2 * Pn)):
        q = qs[y % Pn]
        a[y, :8] = [0, 1, q - 1, q - 2, q - 1, 0, q - 1, 1]
        a[y, n - 4:] = [q - 1, q - 1, 0, q - 2]
        a[y, 1000:1000 + 64] = q - 1
    d_a, d_b = dev(native, a), dev(native, b)
    ctx.forward_batch(d_a, num)
    A = host(native, d_a).reshape(num, n)
    sample = sorted(set(list(range(min(num, 2 * Pn))) + [num // 2, num - 2, num - 1] + ([255, 256, 257] if num > 257 else [])))
    for y in sample:
        assert np.array_equal(A[y], oracle.forward(a[y], prm, y % Pn)), (form, y)
    ctx.forward_batch(d_b, num)
    B = host(native, d_b).reshape(num, n)
    ctx.inverse_batch(d_a, num)
    assert np.array_equal(host(native, d_a).reshape(num, n), a)
    d_f = dev(native, a)
    ctx.polymul_batch(d_f, d_b, num)
    F = host(native, d_f).reshape(num, n)
    for y in sample:
        one = oracle.Params(n, [qs[y % Pn]], [psis[y % Pn]], tables=False)
        want = oracle.inverse(oracle.pointwise_batch(A[y], B[y], one).reshape(-1), prm, y % Pn)
        assert np.array_equal(F[y], want), (form, y)
    ctx.close()


@pytest.mark.parametrize("n", [2048, 4096, 8192, 16384])
@pytest.mark.parametrize("form", sorted(KERNEL_FORMS))
def test_every_headroom_class_at_small_n_matches_oracle(native, oracle, gpu, n, form):
    """the n = 2^11..2^14 kernels are instantiated per headroom class as well: 55- to 62-bit moduli, near 2^k and general"""
    qs, psis32k = KERNEL_FORMS[form]
    psis = [pow(psi, 32768 // n, q) for psi, q in zip(psis32k, qs)]
    Pn, num = len(qs), 3 * len(qs) + 2
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    a = oracle.synth_batch(n, num, qs, 99).reshape(num, n)
    b = oracle.synth_batch(n, num, qs, 98).reshape(num, n)
    for y in range(num):
        q = qs[y % Pn]
        a[y, :4] = [0, q - 1, q - 2, 1]
        a[y, n // 2: n // 2 + 32] = q - 1
    A, B = oracle.forward_batch(a, prm), oracle.forward_batch(b, prm)
    d_a, d_b = dev(native, a), dev(native, b)
    ctx.forward_batch(d_a, num)
    assert np.array_equal(host(native, d_a).reshape(-1), A.reshape(-1))
    ctx.forward_batch(d_b, num)
    d_f = dev(native, a)
    ctx.polymul_batch(d_f, d_b, num)
    assert np.array_equal(host(native, d_f).reshape(-1), oracle.inverse_batch(oracle.pointwise_batch(A, B, prm), prm).reshape(-1))
    ctx.inverse_batch(d_a, num)
    assert np.array_equal(host(native, d_a).reshape(-1), a.reshape(-1))
    ctx.close()


def test_streams_are_respected(native, oracle, gpu):
    """All entry points are asynchronous on the caller's stream (the reference's batch launchers use stream 0)."""
    import torch
    n, qs, psis = 32768, P.Q60, P.PSI60
    ctx = native.NTTContext(n, qs, psis)
    prm = oracle.Params(n, qs, psis)
    a = oracle.synth_batch(n, 8, qs, 31)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d_a = dev(native, a)
        ctx.forward_batch(d_a, 8, stream=s)
        ctx.inverse_batch(d_a, 8, stream=s)
        ctx.forward_batch(d_a, 8, stream=s)
    s.synchronize()
    assert np.array_equal(native.to_host(d_a), oracle.forward_batch(a, prm))
    ctx.close()
