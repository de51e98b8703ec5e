"""GPU: the reference-signature (raw) entry points on the throughput kernels, whole-batch parity at the bench
configuration, logical shards on one device, the literal n = 2^15 path.  Integer work: every word must match."""
import hashlib
import os

import numpy as np
import pytest

import params as P

pytestmark = pytest.mark.gpu

THREADS = os.cpu_count() or 1


def dev(native, a):
    return native.to_device(a)


def host(native, t):
    import torch
    torch.cuda.synchronize()
    return native.to_host(t)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).astype("<u8").tobytes()).hexdigest()


# ------------------------------------------------------------------------------ raw API -> throughput kernels
def test_raw_batch_api_runs_fast_kernels_and_matches_oracle(native, oracle, gpu):
    """forwardNTT_batch / inverseNTT_batch (ntt_60bit.cuh:608,652) with the caller's tables and moduli at the bench
    configuration (n = 32768, 4 x 60-bit, 1024 polynomials): routed to the throughput kernels, and every word equals
    the context API and the CPU oracle."""
    import torch
    n, qs, psis, num = 32768, P.Q60, P.PSI60, 1024
    prm = oracle.Params(n, qs, psis)
    mod = native.Moduli(qs)
    d_tp, d_ti = dev(native, prm.psi_tabs), dev(native, prm.psiinv_tabs)
    native.raw_cache_clear()
    assert native.raw_uses_fast_kernels(n, d_tp, mod) and native.raw_uses_fast_kernels(n, d_ti, mod, inverse=True)
    a = oracle.synth_batch(n, num, qs, 7)
    want = oracle.forward_batch(a, prm, threads=THREADS)
    d_a = dev(native, a)
    native.forwardNTT_batch(d_a, n, d_tp, num, 4, mod)
    got = host(native, d_a)
    assert np.array_equal(got, want)
    ctx = native.NTTContext(n, qs, psis)
    d_b = dev(native, a)
    ctx.forward_batch(d_b, num)
    assert torch.equal(d_a, d_b)
    native.inverseNTT_batch(d_a, n, d_ti, num, 4, mod)
    assert np.array_equal(host(native, d_a), a)
    # the single-polynomial launchers (forwardNTT / inverseNTT, :314,350) take the same route: reference getParams set
    q, psi, psiinv, ninv, qbit = native.getParams(n)
    tp, ti = native.fillTablePsi128(psi, q, psiinv, n)
    d_p, d_i = dev(native, tp), dev(native, ti)
    one = native.Moduli([q])
    assert native.raw_uses_fast_kernels(n, d_p, one)
    x = oracle.splitmix(n, 3, q)
    d_x = dev(native, x)
    native.forwardNTT(d_x, n, None, q, native.barrett_mu(q, qbit), qbit, d_p)
    assert np.array_equal(host(native, d_x), oracle.forward(x, oracle.Params(n, [q], [psi])))
    native.inverseNTT(d_x, n, None, q, native.barrett_mu(q, qbit), qbit, d_i)
    assert np.array_equal(host(native, d_x), x)
    ctx.close()


def test_raw_api_falls_back_to_literal_kernels(native, oracle, gpu):
    """A caller-modified mu, a table that is not psi^bitrev(i), or a Barrett-inexact modulus must NOT take the derived
    context: the literal kernels follow the caller's numbers, and so does the oracle."""
    n, qs, psis, num = 32768, P.Q60, P.PSI60, 64
    prm = oracle.Params(n, qs, psis)
    d_tp, d_ti = dev(native, prm.psi_tabs), dev(native, prm.psiinv_tabs)
    a = oracle.synth_batch(n, num, qs, 11)
    # (1) mu one too small: Algorithm 7 with that mu (results may leave [0, q): both sides compute the same words)
    mu2 = prm.mu.copy()
    mu2[1] -= 1
    mod2 = native.Moduli(qs, mu=mu2, bits=prm.k)
    assert not native.raw_uses_fast_kernels(n, d_tp, mod2)
    prm2 = oracle.Params(n, qs, psis)
    prm2.mu[:] = mu2
    d_a = dev(native, a)
    native.forwardNTT_batch(d_a, n, d_tp, num, 4, mod2)
    assert np.array_equal(host(native, d_a), oracle.forward_batch(a, prm2, threads=THREADS))
    native.inverseNTT_batch(d_a, n, d_ti, num, 4, mod2)
    assert np.array_equal(host(native, d_a), oracle.inverse_batch(oracle.forward_batch(a, prm2, threads=THREADS), prm2, threads=THREADS))
    # (2) a table REWRITTEN IN PLACE after the library has cached a context for its address (the same happens when an
    # allocator hands a freed table's address out again): the per-call comparison must send the call to the literal
    # kernels, which follow the caller's table -- and so does the oracle
    import torch
    mod = native.Moduli(qs)
    d_t3 = dev(native, prm.psi_tabs)
    assert native.raw_uses_fast_kernels(n, d_t3, mod)
    d_a = dev(native, a)
    native.forwardNTT_batch(d_a, n, d_t3, num, 4, mod)
    assert np.array_equal(host(native, d_a), oracle.forward_batch(a, prm, threads=THREADS))
    tabs3 = prm.psi_tabs.copy()
    tabs3[2, 12345] = (int(tabs3[2, 12345]) + 1) % qs[2]
    d_t3.copy_(dev(native, tabs3))                         # same address, other contents
    prm3 = oracle.Params(n, qs, psis)
    prm3.psi_tabs[:] = tabs3
    d_a = dev(native, a)
    native.forwardNTT_batch(d_a, n, d_t3, num, 4, mod)
    assert np.array_equal(host(native, d_a), oracle.forward_batch(a, prm3, threads=THREADS))
    d_t3.copy_(dev(native, prm.psi_tabs))                  # and back: the throughput kernels again, same words as before
    d_a = dev(native, a)
    native.forwardNTT_batch(d_a, n, d_t3, num, 4, mod)
    assert np.array_equal(host(native, d_a), oracle.forward_batch(a, prm, threads=THREADS))
    # a trusted table skips the comparison (the caller's promise); results unchanged
    assert native.raw_trust_tables(n, d_tp, mod) and native.raw_trust_tables(n, d_ti, mod, inverse=True)
    d_a = dev(native, a)
    native.forwardNTT_batch(d_a, n, d_tp, num, 4, mod)
    assert np.array_equal(host(native, d_a), oracle.forward_batch(a, prm, threads=THREADS))
    native.inverseNTT_batch(d_a, n, d_ti, num, 4, mod)
    assert np.array_equal(host(native, d_a), a)
    assert not native.raw_trust_tables(n, d_tp, mod2)      # nothing to trust: a hand-made mu stays on the literal kernels
    # (3) the untouched table still routes to the throughput kernels.  A set that MIXES moduli on which the reference's Barrett is
    # exact with one on which it is not -- the reference's own decryption_test.cu:47-48 set, prime 1 -- is routed per prime since
    # round 5 (the derived context is a mixed one: two of three primes on the lazy butterflies, one on the literal ones)
    assert native.raw_uses_fast_kernels(n, d_tp, mod)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat1_decryption_n4096.npz"))
    kn, kq = int(z["n"]), [int(x) for x in z["q"]]
    kp = oracle.Params(kn, kq, z["psi"])
    d_kt, d_kti, kmod = dev(native, kp.psi_tabs), dev(native, kp.psiinv_tabs), native.Moduli(z["q"])
    assert native.raw_uses_fast_kernels(kn, d_kt, kmod)
    one = oracle.Params(kn, kq[1:2], [int(z["psi"][1])])
    # (round 6: a call whose moduli are ALL inexact runs the single-pass kernels too -- kernel class 0, the reference's own butterflies)
    d_one = dev(native, one.psi_tabs)
    assert native.raw_uses_fast_kernels(kn, d_one, native.Moduli(kq[1:2]))
    oa = oracle.synth_batch(kn, 5, kq[1:2], 31)
    d_oa = dev(native, oa)
    native.forwardNTT_batch(d_oa, kn, d_one, 5, 1, native.Moduli(kq[1:2]))
    assert np.array_equal(host(native, d_oa), oracle.forward_batch(oa, one, threads=THREADS))
    # ... with the reference's words for every prime: checked calls, a table rewritten in place under them (both passes of the class-0
    # kernel must stand back), ragged batches, trusted calls
    for knum in (3, 7, 902):
        ka = oracle.synth_batch(kn, knum, kq, 4100 + knum)
        d_ka = dev(native, ka)
        native.forwardNTT_batch(d_ka, kn, d_kt, knum, 3, kmod)
        want = oracle.forward_batch(ka, kp, threads=THREADS)
        assert np.array_equal(host(native, d_ka), want), knum
        native.inverseNTT_batch(d_ka, kn, d_kti, knum, 3, kmod)
        assert np.array_equal(host(native, d_ka), oracle.inverse_batch(want, kp, threads=THREADS)), knum
    other = oracle.Params(kn, kq, [pow(int(p_), 3, q_) for p_, q_ in zip(z["psi"], kq)])      # psi^3: other primitive roots, other tables
    keep = kp.psi_tabs.copy()
    d_kt.copy_(dev(native, other.psi_tabs))                # same address, other contents: every prime must follow the NEW table
    ka = oracle.synth_batch(kn, 10, kq, 77)
    d_ka = dev(native, ka)
    native.forwardNTT_batch(d_ka, kn, d_kt, 10, 3, kmod)
    assert np.array_equal(host(native, d_ka), oracle.forward_batch(ka, other, threads=THREADS))
    d_kt.copy_(dev(native, keep))
    assert native.raw_trust_tables(kn, d_kt, kmod)
    d_ka = dev(native, ka)
    native.forwardNTT_batch(d_ka, kn, d_kt, 10, 3, kmod)
    assert np.array_equal(host(native, d_ka), oracle.forward_batch(ka, kp, threads=THREADS))


@pytest.mark.parametrize("n", [32768, 65536])
def test_literal_path_large_n_matches_oracle(native, oracle, gpu, n):
    """The literal kernels at n >= 2^15 (one or two stage launches + the 2^14-coefficient LDS kernel): the reference's
    words, checked against the oracle with a Barrett-exact prime set forced onto the literal path by a table tweak that
    the transforms never read (entry 0)."""
    qs = [P.EDGE_PRIMES[59][0], P.EDGE_PRIMES[61][0]]
    psis = [P.EDGE_PRIMES[59][1][n], P.EDGE_PRIMES[61][1][n]]
    prm = oracle.Params(n, qs, psis)
    num = 10
    a = oracle.synth_batch(n, num, qs, 5)
    # a hand-made mu for prime 0 keeps the call on the literal kernels
    mu = prm.mu.copy()
    mu[0] -= 2
    prm.mu[:] = mu
    mod = native.Moduli(qs, mu=mu, bits=prm.k)
    d_tp, d_ti = dev(native, prm.psi_tabs), dev(native, prm.psiinv_tabs)
    assert not native.raw_uses_fast_kernels(n, d_tp, mod)
    d_a = dev(native, a)
    native.forwardNTT_batch(d_a, n, d_tp, num, 2, mod)
    A = oracle.forward_batch(a, prm, threads=THREADS)
    assert np.array_equal(host(native, d_a), A)
    native.inverseNTT_batch(d_a, n, d_ti, num, 2, mod)
    assert np.array_equal(host(native, d_a), oracle.inverse_batch(A, prm, threads=THREADS))


def test_raw_barrett_large_batches(native, oracle, gpu):
    """barrett_batch_3param over more polynomials than a grid's y extent (65535) and on an unaligned view."""
    n, qs, psis = 2048, [P.REF_PARAMS[2048][0], P.Q55[0]], None
    prm = oracle.Params(n, qs, [P.REF_PARAMS[2048][1], 1], tables=False)
    mod = native.Moduli(qs)
    num = 66000
    rng = np.random.default_rng(5)
    a = (rng.integers(0, 1 << 62, size=(num, n), dtype=np.uint64) % np.array(qs, dtype=np.uint64)[np.arange(num) % 2][:, None])
    b = (rng.integers(0, 1 << 62, size=(num, n), dtype=np.uint64) % np.array(qs, dtype=np.uint64)[np.arange(num) % 2][:, None])
    d_a, d_b = dev(native, a), dev(native, b)
    d_c = dev(native, np.zeros_like(a))
    native.barrett_batch_3param(d_c, d_a, d_b, n, 2, mod)
    want = oracle.pointwise_batch(a, b, prm, division=2)
    assert np.array_equal(host(native, d_c), want)
    # 8-byte aligned only: the scalar form of the kernel
    import torch
    flat_a = torch.cat([torch.zeros(1, dtype=torch.int64, device=gpu), d_a.reshape(-1)])[1:1 + 4 * n]
    flat_b = torch.cat([torch.zeros(1, dtype=torch.int64, device=gpu), d_b.reshape(-1)])[1:1 + 4 * n]
    out = torch.zeros(4 * n + 1, dtype=torch.int64, device=gpu)[1:]
    assert flat_a.data_ptr() % 16 == 8
    native.barrett_batch_3param(out, flat_a, flat_b, n, 2, mod, num=4)
    assert np.array_equal(host(native, out).reshape(4, n), want[:4])


# ------------------------------------------------------------------------------ whole batch vs the oracle
@pytest.mark.parametrize("num", [256, 1024])
def test_full_batch_digest_against_oracle(native, oracle, gpu, num):
    """BASELINE configs[2] (256) and configs[3]'s per-GPU shard (1024): SHA-256 over ALL output polynomials of forward,
    the fused product and the inverse against the oracle run with OpenMP over polynomials."""
    n, qs, psis = 32768, P.Q60, P.PSI60
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    a = oracle.synth_batch(n, num, qs, 1000 + num)
    b = oracle.synth_batch(n, num, qs, 5000 + num)
    A = oracle.forward_batch(a, prm, threads=THREADS)
    B = oracle.forward_batch(b, prm, threads=THREADS)
    C = oracle.inverse_batch(oracle.pointwise_batch(A, B, prm), prm, threads=THREADS)
    d_a, d_b = dev(native, a), dev(native, b)
    ctx.forward_batch(d_b, num)
    assert sha(host(native, d_b)) == sha(B)
    ctx.polymul_batch(d_a, d_b, num)
    assert sha(host(native, d_a)) == sha(C)
    ctx.inverse_batch(d_b, num)
    assert sha(host(native, d_b)) == sha(b)
    ctx.close()


# ------------------------------------------------------------------------------ 8 logical shards on one device
@pytest.mark.parametrize("num", [1024, 1023])
def test_eight_logical_shards_equal_whole_batch(native, oracle, gpu, num):
    """the N-GPU decomposition (shard.shard_range: whole polynomials, shard starts at multiples of the
    prime count) executed as 8 shards on cuda:0 gives the words of the whole-batch call, for forward, inverse and the
    fused product, also with a ragged tail."""
    import torch
    from ntt_cuda_amd import shard
    n, qs, psis, world = 32768, P.Q60, P.PSI60, 8
    ctx = native.NTTContext(n, qs, psis)
    g = torch.Generator(device=gpu).manual_seed(99 + num)
    qcol = torch.tensor(np.array(qs, dtype=np.uint64).view(np.int64), device=gpu)[torch.arange(num, device=gpu) % 4].unsqueeze(1)
    a = torch.randint(0, 1 << 60, (num, n), dtype=torch.int64, device=gpu, generator=g)
    b = torch.randint(0, 1 << 60, (num, n), dtype=torch.int64, device=gpu, generator=g)
    a = torch.where(a >= qcol, a - qcol, a).contiguous()
    b = torch.where(b >= qcol, b - qcol, b).contiguous()
    whole_f = a.clone()
    ctx.forward_batch(whole_f, num)
    whole_m = a.clone()
    bh = b.clone()
    ctx.forward_batch(bh, num)
    ctx.polymul_batch(whole_m, bh, num)
    covered = 0
    sh_f, sh_m = a.clone(), a.clone()
    for rank in range(world):
        s, c = shard.shard_range(num, 4, rank, world)
        assert s % 4 == 0 and s == covered
        covered += c
        if c == 0:
            continue
        piece = sh_f[s:s + c]                     # a view: shards are contiguous ranges of whole polynomials
        ctx.forward_batch(piece, c)
        ctx.polymul_batch(sh_m[s:s + c], bh[s:s + c], c)
    assert covered == num
    assert torch.equal(sh_f, whole_f) and torch.equal(sh_m, whole_m)
    for rank in range(world):
        s, c = shard.shard_range(num, 4, rank, world)
        if c:
            ctx.inverse_batch(sh_f[s:s + c], c)
    assert torch.equal(sh_f, a)
    # the sample against the oracle (the whole-batch call itself is pinned by test_full_batch_digest_against_oracle)
    prm = oracle.Params(n, qs, psis)
    for y in (0, num // 8 + 1, num - 1):
        assert np.array_equal(native.to_host(whole_f[y].contiguous()), oracle.forward(native.to_host(a[y].contiguous()), prm, y % 4))
    ctx.close()


def test_context_calls_run_on_the_context_device(native, gpu):
    """The C ABI switches to the context's device for the call and back (no silent launch on another device): with one
    GPU this can only check the bookkeeping -- the context reports its device and the caller's device is untouched."""
    import torch
    ctx = native.NTTContext(2048, [P.REF_PARAMS[2048][0]], [P.REF_PARAMS[2048][1]], device=0)
    assert native.lib().mi355ntt_ctx_device(ctx._h) == 0
    before = torch.cuda.current_device()
    x = torch.zeros(2048, dtype=torch.int64, device=gpu)
    ctx.forward(x, 0)
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == before
    with pytest.raises(native.NTTError):
        native.NTTContext(2048, [P.REF_PARAMS[2048][0]], [P.REF_PARAMS[2048][1]], device=63)
    assert torch.cuda.current_device() == before
    ctx.close()


# ------------------------------------------------------------------------------ fused product with shared second operands
@pytest.mark.parametrize("n,qs_name,num,group", [(32768, "Q60", 22, 8), (32768, "Q60", 300, 0), (4096, "Q60", 13, 4), (2048, "Q60", 9, 0),
                                                  (65536, "EDGE", 6, 2), (4096, "KAT", 7, 0)])
def test_polymul_batch_shared_matches_oracle(native, oracle, gpu, n, qs_name, num, group):
    """mi355ntt_polymul_batch_shared: polynomial y multiplies with bhat[(y // group) * division + y % division] (group 0: one
    group).  Against the oracle's forward -> pointwise -> inverse (bfv_encryption.cuh:268-271 with one key for many
    ciphertexts), ragged batches, on the fused kernels (persistent and latency path), the split n = 2^16 composition and
    the literal kernels (KAT-1 moduli)."""
    if qs_name == "Q60":
        qs = P.Q60
        psis = [pow(psi, 32768 // n, q) for psi, q in zip(P.PSI60, qs)]
    elif qs_name == "EDGE":
        qs, psis = [P.EDGE_PRIMES[b][0] for b in (59, 61)], [P.EDGE_PRIMES[b][1][n] for b in (59, 61)]
    else:
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat1_decryption_n4096.npz"))
        qs, psis = [int(x) for x in z["q"]], [int(x) for x in z["psi"]]
    D = len(qs)
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    assert ctx.uses_literal_kernels == (qs_name == "KAT")
    groups = -(-num // group) if group else 1
    a = oracle.synth_batch(n, num, qs, 31)
    keys = oracle.forward_batch(oracle.synth_batch(n, groups * D, qs, 32), prm)                 # NTT-domain operands, prime y % D
    b_full = np.stack([keys[((y // group) if group else 0) * D + y % D] for y in range(num)])
    want = oracle.inverse_batch(oracle.pointwise_batch(oracle.forward_batch(a, prm), b_full, prm), prm)
    d_a = dev(native, a)
    ctx.polymul_batch_shared(d_a, dev(native, keys), num, D, group)
    assert np.array_equal(host(native, d_a), want)
    ctx.close()


def test_polymul_batch_shared_rejects_bad_groups(native, gpu):
    ctx = native.NTTContext(4096, P.Q60, [pow(psi, 8, q) for psi, q in zip(P.PSI60, P.Q60)])
    import torch
    a = torch.zeros((8, 4096), dtype=torch.int64, device=gpu)
    with pytest.raises(native.NTTError):
        ctx.polymul_batch_shared(a, a, 8, 4, 6)          # group not a multiple of division
    ctx.close()


def test_empty_batches_are_no_ops(native, gpu):
    """num = 0 / count = 0 on the round-2 entry points: accepted, nothing launched, nothing touched"""
    import torch
    from ntt_cuda_amd import bfv
    n = 4096
    qs = P.Q60[:3]
    psis = [pow(psi, 32768 // n, q) for psi, q in zip(P.PSI60, qs)]
    ctx = native.NTTContext(n, qs, psis)
    a = torch.full((3, n), 5, dtype=torch.int64, device=gpu)
    ctx.polymul_batch_shared(a, a, 0, 3, 0)
    b = bfv.BFVContext(n, qs, psis, 1024, P.GAMMA61)
    c = torch.full((2 * 3, n), 7, dtype=torch.int64, device=gpu)
    L = native.lib()
    assert L.mi355ntt_bfv_encrypt_batch(b._h, c.data_ptr(), c.data_ptr(), c.data_ptr(), c.data_ptr(), 0, None) == 0
    assert L.mi355ntt_bfv_decrypt_batch(b._h, c.data_ptr(), c.data_ptr(), 0, None) == 0
    torch.cuda.synchronize()
    assert int(a.sum()) == 5 * 3 * n and int(c.sum()) == 7 * 6 * n
    b.close()
    ctx.close()


@pytest.mark.parametrize("primes,num", [(3, 600), (5, 777)])
def test_persistent_walk_with_modulus_count_not_dividing_the_grid(native, oracle, gpu, primes, num):
    """n = 2^15 with more polynomials than workgroups and a prime count that does not divide the grid (256): the persistent
    kernels carry the modulus index incrementally (forward walking up, inverse walking DOWN): every word of a ragged batch
    against the oracle, forward, inverse and fused product."""
    n = 32768
    qs = (P.Q60 + [P.Q60_SPECIAL])[:primes]
    psis = (P.PSI60 + [P.PSI60_SPECIAL])[:primes]
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    a = oracle.synth_batch(n, num, qs, 5)
    A = oracle.forward_batch(a, prm, threads=THREADS)
    d = dev(native, a)
    ctx.forward_batch(d, num)
    assert sha(host(native, d)) == sha(A)
    ctx.inverse_batch(d, num)
    assert np.array_equal(host(native, d), a)
    b = oracle.synth_batch(n, num, qs, 6)
    B = oracle.forward_batch(b, prm, threads=THREADS)
    want = oracle.inverse_batch(oracle.pointwise_batch(A, B, prm), prm, threads=THREADS)
    d_b = dev(native, B)
    ctx.polymul_batch(d, d_b, num)
    assert sha(host(native, d)) == sha(want)
    # inverse alone on NTT-domain data that is not a forward image of anything special
    d2 = dev(native, B)
    ctx.inverse_batch(d2, num)
    assert sha(host(native, d2)) == sha(oracle.inverse_batch(B, prm, threads=THREADS))
    ctx.close()


@pytest.mark.parametrize("n,num", [(2048, 9001), (4096, 5003), (8192, 2101), (16384, 701)])
def test_small_ring_sizes_walk_more_polynomials_than_workgroups(native, oracle, gpu, n, num):
    """n = 2^11..2^14: batches larger than the resident grid (every workgroup walks several polynomials; the rows enter and
    leave through the wave-local staging with the next polynomial's loads behind the stores): whole-batch digests against
    the oracle for forward, inverse and the fused product, 3 primes (does not divide the grid)."""
    qs = P.Q60[:3]
    psis = [pow(psi, 32768 // n, q) for psi, q in zip(P.PSI60, qs)]
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    a = oracle.synth_batch(n, num, qs, 9)
    A = oracle.forward_batch(a, prm, threads=THREADS)
    d = dev(native, a)
    ctx.forward_batch(d, num)
    assert sha(host(native, d)) == sha(A)
    ctx.inverse_batch(d, num)
    assert np.array_equal(host(native, d), a)
    d_b = dev(native, A)
    ctx.polymul_batch(d, d_b, num)
    assert sha(host(native, d)) == sha(oracle.inverse_batch(oracle.pointwise_batch(A, A, prm), prm, threads=THREADS))
    ctx.close()
