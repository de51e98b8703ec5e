"""GPU: the BFV launch layer -- bootstrap constants, keygen_rns / encryption_rns /
decryption_rns after their samplers -- against the oracle's literal restatement and the reference's known-answer test."""
import os

import numpy as np
import pytest

import params as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat1_decryption_n4096.npz")


def demo_moduli(n, count):
    """the reference's 55-bit demo moduli (demo.cu:35-36) with psi lowered to the 2n-th root for ring degree n"""
    qs = P.Q55[:count]
    return qs, [pow(psi, 32768 // n, q) for psi, q in zip(P.PSI55, qs)]


@pytest.mark.gpu
def test_bootstrap_constants_match_oracle(native, oracle, gpu):
    """demo.cu:62-272 on the host vs the oracle restatement (itself pinned on the reference's constants in
    tests/test_oracle_golden.py), for the KAT-1 moduli and the 55-bit demo moduli"""
    from ntt_cuda_amd import bfv
    z = np.load(GOLD)
    for n, qs, psis, t, gamma in [(int(z["n"]), [int(x) for x in z["q"]], [int(x) for x in z["psi"]], int(z["t"]), int(z["gamma"])),
                                  (32768, P.Q55, P.PSI55, 1024, P.GAMMA61)]:
        want = oracle.bfv_constants(qs, psis, t, gamma)
        ctx = bfv.BFVContext(n, qs, psis, t, gamma)
        got = ctx.constants()
        for key in ("inv_punctured_q", "neg_inv_q_mod_t_gamma", "prod_t_gamma_mod_q", "inv_q_last_mod_q", "qi_div_t"):
            assert np.array_equal(got[key], want[key]), key
        ctx.close()


@pytest.mark.gpu
def test_decryption_replays_reference_kat(native, oracle, gpu):
    """decryption_test.cu:348,355: the embedded ciphertext decrypts to m[i] = i % 10, and every word the reference's
    decryption_rns leaves in c is reproduced (oracle = literal restatement of all its kernels)."""
    import torch
    from ntt_cuda_amd import bfv
    z = np.load(GOLD)
    n = int(z["n"])
    qs, psis = [int(x) for x in z["q"]], [int(x) for x in z["psi"]]
    ctx = bfv.BFVContext(n, qs, psis, int(z["t"]), int(z["gamma"]))
    assert ctx.uses_literal_kernels                  # the second KAT modulus is Barrett-inexact (test_barrett_exactness.py)
    c = native.to_device(z["c_host"])
    sk = native.to_device(z["sk_host"])
    m = ctx.decrypt(c, sk)
    torch.cuda.synchronize()
    assert np.array_equal(native.to_host(m), np.arange(n, dtype=np.uint64) % 10)
    want_c = z["c_host"].copy()
    oracle.lib().orc_bfv_decrypt(oracle._p(want_c), oracle._p(z["sk_host"]), oracle._p(np.array(qs, np.uint64)), oracle._p(np.array(psis, np.uint64)),
                                 len(qs), n, int(z["t"]), int(z["gamma"]), oracle._p(np.empty(n, np.uint64)), None)
    assert np.array_equal(native.to_host(c), want_c)
    ctx.close()
    # the exact kernels decrypt the same ciphertext to the same plaintext
    ctx = bfv.BFVContext(n, qs, psis, int(z["t"]), int(z["gamma"]), exact_on_inexact_primes=True)
    c = native.to_device(z["c_host"])
    assert np.array_equal(native.to_host(ctx.decrypt(c, sk)), np.arange(n, dtype=np.uint64) % 10)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,count", [(4096, 3), (32768, 4), (32768, 60)])
def test_keygen_encrypt_decrypt_match_oracle_and_round_trip(native, oracle, gpu, n, count):
    """Same sampled inputs through the GPU drivers and the oracle: every output word equal; and the message comes back
    (demo.cu:302-311)."""
    import torch
    from ntt_cuda_amd import bfv
    if count == 60:      # BASELINE configs[4]: four 60-bit primes + the special one
        qs, psis = P.Q60 + [P.Q60_SPECIAL], P.PSI60 + [P.PSI60_SPECIAL]
    else:
        qs, psis = demo_moduli(n, count)
    t, gamma = 1024, P.GAMMA61
    R = len(qs)
    ctx = bfv.BFVContext(n, qs, psis, t, gamma)
    assert not ctx.uses_literal_kernels
    smp = oracle.bfv_sample(qs, n, 11)
    pk = np.zeros((2, R, n), dtype=np.uint64)
    pk[1] = smp["uniform"]
    e_k = smp["err"]()
    want_sk, want_pk = oracle.bfv_keygen_core(smp["ternary"], pk, e_k, qs, psis, n)
    d_sk, d_pk = native.to_device(smp["ternary"]), native.to_device(pk)
    ctx.keygen(d_sk, d_pk, native.to_device(e_k))
    torch.cuda.synchronize()
    assert np.array_equal(native.to_host(d_sk).reshape(R, n), want_sk)
    assert np.array_equal(native.to_host(d_pk).reshape(2, R, n), want_pk)

    m = smp["rng"].integers(0, t, size=n, dtype=np.uint64)
    u = oracle.bfv_sample(qs, n, 111)["ternary"]
    e = np.stack([smp["err"](), smp["err"]()])
    c0 = np.stack([u, u])
    want_c = oracle.bfv_encrypt_core(c0, want_pk, e, m, qs, psis, n, t)
    d_c = native.to_device(c0)
    ctx.encrypt(d_c, d_pk, native.to_device(e), native.to_device(m))
    torch.cuda.synchronize()
    assert np.array_equal(native.to_host(d_c).reshape(2, R, n), want_c.reshape(2, R, n))

    want_after = want_c.reshape(-1).copy()
    want_m = np.empty(n, np.uint64)
    oracle.lib().orc_bfv_decrypt(oracle._p(want_after), oracle._p(np.ascontiguousarray(want_sk.reshape(-1)[: (R - 1) * n])),
                                 oracle._p(np.array(qs, np.uint64)), oracle._p(np.array(psis, np.uint64)), R, n, t, gamma,
                                 oracle._p(want_m), None)
    got_m = ctx.decrypt(d_c, d_sk)
    torch.cuda.synchronize()
    assert np.array_equal(native.to_host(got_m), m) and np.array_equal(want_m, m)
    assert np.array_equal(native.to_host(d_c).reshape(-1), want_after)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gamma", [P.GAMMA40, (1 << 50) - 27, P.GAMMA61])
def test_decrypt_with_moduli_wider_than_gamma_matches_oracle(native, oracle, gpu, gamma):
    """62-bit q_i next to a 40- / 50-bit gamma.  The base conversion's Barrett products mod gamma then exceed 2 gamma
    (operands of 62 bits on a 40-bit Barrett), and a lazily summed accumulator would wrap where the reference's per-term `% gamma`
    (poly_arithmetic.cuh:252) does not: every word decryption_rns leaves behind must still equal the oracle's, single and batched."""
    import torch
    from ntt_cuda_amd import bfv
    n, t = 4096, 1024
    qs, psis = [q for q, _ in P.Q62_N4096], [w for _, w in P.Q62_N4096]
    R = len(qs)
    ctx = bfv.BFVContext(n, qs, psis, t, gamma)
    rng = np.random.default_rng(2024)
    count = 3
    c = np.stack([np.stack([np.stack([rng.integers(0, q, size=n, dtype=np.uint64) for q in qs]) for _ in range(count)]) for _ in range(2)])   # [2][count][R][n]
    sk_hat = oracle.bfv_sample(qs, n, 9)["uniform"]      # (the drivers take the key in the NTT domain: any residues serve)
    want = []
    for z_ in range(count):
        w_c = np.ascontiguousarray(c[:, z_]).reshape(-1).copy()
        w_m = np.empty(n, np.uint64)
        oracle.lib().orc_bfv_decrypt(oracle._p(w_c), oracle._p(np.ascontiguousarray(sk_hat.reshape(-1)[: (R - 1) * n])), oracle._p(np.array(qs, np.uint64)),
                                     oracle._p(np.array(psis, np.uint64)), R, n, t, gamma, oracle._p(w_m), None)
        want.append((w_c, w_m))
    d_sk = native.to_device(sk_hat)
    for z_ in range(count):
        d_c = native.to_device(np.ascontiguousarray(c[:, z_]))
        got_m = ctx.decrypt(d_c, d_sk)
        torch.cuda.synchronize()
        assert np.array_equal(native.to_host(got_m), want[z_][1]), "plaintext words, ciphertext %d" % z_
        assert np.array_equal(native.to_host(d_c).reshape(-1), want[z_][0]), "ciphertext words, ciphertext %d" % z_
    d_cb = native.to_device(c)
    ctx.decrypt_batch(d_cb, d_sk, count)
    torch.cuda.synchronize()
    got = native.to_host(d_cb).reshape(2, count, R, n)
    for z_ in range(count):
        w_ = want[z_][0].reshape(2, R, n)
        assert np.array_equal(got[0, z_], w_[0]), "batched driver, ciphertext %d" % z_
        assert np.array_equal(got[1, z_, : R - 1], w_[1, : R - 1]), "batched driver, ciphertext %d" % z_      # (the dropped prime's c1 slot is scratch in the batch)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,nprimes,count,literal", [(4096, 3, 5, False), (32768, 4, 7, False), (32768, 60, 64, False), (4096, 0, 3, True),
                                                     (65536, 2, 3, False), (32768, 60, 52, False), (32768, 60, 110, False)])
def test_batched_drivers_equal_looped_single_calls(native, oracle, gpu, n, nprimes, count, literal):
    """mi355ntt_bfv_encrypt_batch / _decrypt_batch over `count` ciphertexts (layout [2][count][R][n]) leave, for each
    ciphertext, exactly the words the single drivers leave (those are pinned on the oracle above) -- on the fused
    product kernels, on the three-step composition (n = 65536) and on the literal kernels (KAT-1 moduli).  64, 52 and 110 ciphertexts on
    4 + 1 primes: 320 / 260 / 550 polynomials per product = one or two full rounds of the persistent grid plus a short tail -- the head
    on the persistent kernel, the tail on the small-batch kernels, both with the decryption's scaling step in their store path."""
    import torch
    from ntt_cuda_amd import bfv
    if literal:
        z = np.load(GOLD)
        qs, psis, t, gamma = [int(x) for x in z["q"]], [int(x) for x in z["psi"]], int(z["t"]), int(z["gamma"])
    elif nprimes == 60:
        qs, psis, t, gamma = P.Q60 + [P.Q60_SPECIAL], P.PSI60 + [P.PSI60_SPECIAL], 1024, P.GAMMA61
    elif n == 65536:
        qs, psis, t, gamma = [P.EDGE_PRIMES[b][0] for b in (59, 61)], [P.EDGE_PRIMES[b][1][65536] for b in (59, 61)], 1024, P.GAMMA61
    else:
        (qs, psis), t, gamma = demo_moduli(n, nprimes), 1024, P.GAMMA61
    R = len(qs)
    ctx = bfv.BFVContext(n, qs, psis, t, gamma)
    assert ctx.uses_literal_kernels == literal
    smp = oracle.bfv_sample(qs, n, 5)
    pk = np.zeros((2, R, n), dtype=np.uint64)
    pk[1] = smp["uniform"]
    d_sk, d_pk = native.to_device(smp["ternary"]), native.to_device(pk)
    ctx.keygen(d_sk, d_pk, native.to_device(smp["err"]()))
    rng = np.random.default_rng(77)
    m = rng.integers(0, t, size=(count, n), dtype=np.uint64)
    u = np.stack([oracle.bfv_sample(qs, n, 1000 + z_)["ternary"] for z_ in range(count)])          # [count][R][n]
    e = np.stack([np.stack([smp["err"]() for _ in range(count)]) for _ in range(2)])                # [2][count][R][n]
    c_batch = native.to_device(np.stack([u, u]))
    ctx.encrypt_batch(c_batch, d_pk, native.to_device(e), native.to_device(m), count)
    torch.cuda.synchronize()
    got = native.to_host(c_batch).reshape(2, count, R, n)
    singles = []
    for z_ in range(count):
        d_c = native.to_device(np.stack([u[z_], u[z_]]))
        ctx.encrypt(d_c, d_pk, native.to_device(np.ascontiguousarray(e[:, z_])), native.to_device(m[z_]))
        singles.append(d_c)
        assert np.array_equal(native.to_host(d_c).reshape(2, R, n), got[:, z_]), z_
    ctx.decrypt_batch(c_batch, d_sk, count)
    torch.cuda.synchronize()
    got = native.to_host(c_batch).reshape(2, count, R, n)
    for z_ in range(count):
        ctx.decrypt(singles[z_], d_sk)
        want = native.to_host(singles[z_]).reshape(2, R, n)
        assert np.array_equal(got[0, z_], want[0]), z_
        assert np.array_equal(got[1, z_, : R - 1], want[1, : R - 1]), z_       # the dropped prime's c1 slot is scratch in the batch
        assert np.array_equal(got[0, z_, R - 2], m[z_]), z_                     # and the message is back
    ctx.close()


def test_bfv_create_rejects_unsupported_parameters(native):
    """host-side validation happens before any GPU call"""
    import ctypes
    L = native.lib()
    h = native.vp()
    qs, psis = demo_moduli(4096, 3)
    a = lambda v: np.array(v, np.uint64).ctypes.data_as(native.u64p)
    assert L.mi355ntt_bfv_create(ctypes.byref(h), 4096, 3, a(qs), a(psis), 1000, P.GAMMA61, 0, 0) == native.EUNSUPPORTED   # t not 2^k
    assert L.mi355ntt_bfv_create(ctypes.byref(h), 4096, 1, a(qs), a(psis), 1024, P.GAMMA61, 0, 0) == native.EUNSUPPORTED   # no special prime
    assert L.mi355ntt_bfv_create(ctypes.byref(h), 4096, 3, a(qs), a(psis), 1024, 1 << 61, 0, 0) == native.EUNSUPPORTED     # gamma even
    assert L.mi355ntt_bfv_create(ctypes.byref(h), 4096, 3, a(qs), a(psis), 1 << 20, P.GAMMA61, 0, 0) == native.EPARAM      # q != 1 mod t
