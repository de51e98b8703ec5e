"""CPU: pin the oracle against every known-answer vector the reference holds for the NTT hot path
 and against the algebraic identities of the transform."""
import os

import numpy as np
import pytest

import params as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat1_decryption_n4096.npz")


@pytest.fixture(scope="module")
def kat():
    return np.load(GOLD)


def test_kat1_fixture_digests(kat):
    # digests of the arrays embedded at decryption_test.cu:348,355
    assert P.digest(kat["c_host"]) == "5867af3e9134a636156e7993a789ef490981a3b40f8b4aa051c2c9757cc6519b"
    assert P.digest(kat["sk_host"]) == "b731d60ed7b47052341bb7bb059749268d2100e3d2817054b94e46d80f11eb05"


def test_kat1_decryption_yields_reference_plaintext(oracle, kat):
    """decryption_test.cu:230-232,380-388: the embedded ciphertext decrypts to m[i] = i % 10."""
    n = int(kat["n"])
    out, stages = oracle.bfv_decrypt(kat["c_host"], kat["sk_host"], kat["q"], kat["psi"], n, int(kat["t"]), int(kat["gamma"]),
                                     want_stages=True)
    assert np.array_equal(out, np.arange(n, dtype=np.uint64) % 10)
    for got, want in zip(stages, P.KAT1_STAGE_DIGESTS):
        assert P.digest(got) == want
    assert list(stages[0][:2]) == [63382629086, 64841577898]
    assert list(stages[2][:2]) == [989152337, 61837867373]


def test_known_answer_constants(oracle, kat):
    """old/decryption.cu:46,97,103,113 and old/encryption.cu:98,101 list these derived constants."""
    c = oracle.bfv_constants(kat["q"], kat["psi"], int(kat["t"]), int(kat["gamma"]))
    assert list(c["psiinv"][:2]) == [60243494989, 30331733829]
    assert list(c["inv_punctured_q"]) == [26179219651, 42540076863]
    assert list(c["neg_inv_q_mod_t_gamma"]) == [1023, 803320262470649134]
    assert list(c["prod_t_gamma_mod_q"]) == [37067052033, 64547873793]
    assert list(c["inv_q_last_mod_q"]) == [20955999355, 17095778744]
    assert list(c["qi_div_t"]) == [67108792, 67108624, 134217600]
    # decryption_test.cu Barrett constants as the survey recorded them
    prm = oracle.Params(4096, kat["q"][:2], kat["psi"][:2], tables=False)
    assert list(prm.mu) == [68719550463, 68719722495] and list(prm.k) == [36, 36]


@pytest.mark.parametrize("n", sorted(P.REF_PARAMS))
def test_parameter_h_tuples(oracle, n):
    """parameter.h:31-79: psiinv, ninv and q_bit are consistent with q, psi under the restated helpers."""
    q, psi, psiinv, ninv, qbit = P.REF_PARAMS[n]
    L = oracle.lib()
    assert L.orc_modinv(psi, q) == psiinv
    assert L.orc_modinv(n, q) == ninv
    assert L.orc_bit_length(q) == qbit
    assert L.orc_modpow(psi, n, q) == q - 1          # primitive 2n-th root
    got = [oracle.u64() for _ in range(4)]
    import ctypes
    bits = ctypes.c_uint()
    assert L.orc_get_params(n, *[ctypes.byref(g) for g in got], ctypes.byref(bits)) == 0
    assert [g.value for g in got] + [bits.value] == [q, psi, psiinv, ninv, qbit]


def test_parameter_h_58bit_set(oracle):
    q, psi, psiinv, ninv, qbit = P.REF_PARAMS_4096_58BIT
    L = oracle.lib()
    assert (L.orc_modinv(psi, q), L.orc_modinv(4096, q), L.orc_bit_length(q)) == (psiinv, ninv, qbit)


@pytest.mark.parametrize("key", sorted(P.GOLDEN_DIGESTS))
def test_survey_golden_digests(oracle, key):
    """table, NTT(a) and negacyclic product digests for the BASELINE configs."""
    n, q, psi = key
    d_tab, d_ntt, d_mul = P.GOLDEN_DIGESTS[key]
    prm = oracle.Params(n, [q], [psi])
    a, b = oracle.splitmix(n, 1, q), oracle.splitmix(n, 2, q)
    assert P.digest(prm.psi_tabs[0]) == d_tab
    A, B = oracle.forward(a, prm), oracle.forward(b, prm)
    assert P.digest(A) == d_ntt
    C = oracle.inverse(oracle.pointwise_batch(A, B, prm), prm)
    assert P.digest(C) == d_mul
    assert np.array_equal(oracle.inverse(A, prm), a)


def test_barrett_matches_int128_mod(oracle):
    """Algorithm 7 (ntt_60bit.cuh:44-61) equals a*b mod q for canonical inputs, all moduli sizes."""
    rng = np.random.default_rng(7)
    L = oracle.lib()
    moduli = [P.REF_PARAMS[n][0] for n in P.REF_PARAMS] + P.Q60 + P.Q55 + [v[0] for v in P.EDGE_PRIMES.values()] + [P.GAMMA61]
    for q in moduli:
        k = L.orc_bit_length(q)
        mu = L.orc_mu(q, k)
        xs = [0, 1, q - 1, q - 2, q // 2, (1 << (k - 1)) % q] + [int(x) % q for x in rng.integers(0, 1 << 63, 200, dtype=np.uint64)]
        for a in xs[:40]:
            for b in (xs[-1], q - 1, 1, xs[17]):
                assert L.orc_barrett(a, b, q, mu, k) == (a * b) % q


@pytest.mark.parametrize("n,q,psi", [(256, 1073479681, None), (1024, 576460752300015617, None), (2048, 137438691329, 22157790)])
def test_polymul_matches_reference_schoolbook(oracle, n, q, psi):
    """60bit_ntt_test.cu:65-66,85-98 (the reference's own, normally disabled, check): INTT(NTT(a).NTT(b))
    equals refPolyMul128 (helper.h:95-126)."""
    if psi is None:
        psi = next(pow(x, (q - 1) // (2 * n), q) for x in range(2, 100) if pow(pow(x, (q - 1) // (2 * n), q), n, q) == q - 1)
    prm = oracle.Params(n, [q], [psi])
    a, b = oracle.splitmix(n, 11, q), oracle.splitmix(n, 12, q)
    got = oracle.inverse(oracle.pointwise_batch(oracle.forward(a, prm), oracle.forward(b, prm), prm), prm)
    assert np.array_equal(got, oracle.ref_polymul(a, b, q))


def test_forward_is_the_negacyclic_evaluation(oracle):
    """forward(a)[bitrev(i)] = sum_j a_j psi^((2i+1) j) mod q."""
    n, (q, roots) = 64, P.EDGE_PRIMES[30]
    psi = pow(roots[4096], 4096 // n, q)
    prm = oracle.Params(n, [q], [psi])
    a = oracle.splitmix(n, 3, q)
    A = oracle.forward(a, prm)
    L = oracle.lib()
    for i in range(n):
        want = sum(int(a[j]) * pow(psi, (2 * i + 1) * j, q) for j in range(n)) % q
        assert int(A[L.orc_bitrev(i, 6)]) == want


def test_batch_index_rules(oracle):
    """ntt_60bit.cuh:391-422: polynomial y uses modulus/table y % division."""
    n = 4096
    qs = [P.REF_PARAMS_4096_58BIT[0], P.REF_PARAMS[4096][0], P.EDGE_PRIMES[59][0]]
    psis = [P.REF_PARAMS_4096_58BIT[1], P.REF_PARAMS[4096][1], P.EDGE_PRIMES[59][1][4096]]
    prm = oracle.Params(n, qs, psis)
    x = oracle.synth_batch(n, 5, qs)              # ragged: 5 polynomials over 3 primes
    F = oracle.forward_batch(x, prm, division=3).reshape(5, n)
    for y in range(5):
        single = oracle.Params(n, [qs[y % 3]], [psis[y % 3]])
        assert np.array_equal(F[y], oracle.forward(x[y], single))
    assert np.array_equal(oracle.inverse_batch(F, prm, division=3).reshape(5, n), x)
    # OpenMP batch == serial batch
    assert np.array_equal(oracle.forward_batch(x, prm, division=3, threads=4).reshape(5, n), F)


def test_bfv_keygen_encrypt_decrypt_round_trip(oracle):
    """demo.cu:302-311: the only check the reference has for keygen_rns / encryption_rns is that decryption returns the
    message.  Same here for the restatements (decryption itself is pinned by KAT-1 above), on three of the reference's
    55-bit demo moduli (demo.cu:35-36) at n = 4096.  (Not on the KAT-1 moduli: the second of them makes the reference's
    Barrett inexact, and with ternary inputs its own round trip breaks -- tests/test_barrett_exactness.py.)"""
    n, t, gamma = 4096, 1024, P.GAMMA61
    qs = P.Q55[:3]
    psis = [pow(psi, 32768 // n, q) for psi, q in zip(P.PSI55, qs)]
    R = len(qs)
    for seed in (1, 2, 3):
        smp = oracle.bfv_sample(qs, n, seed)
        pk = np.zeros((2, R, n), dtype=np.uint64)
        pk[1] = smp["uniform"]
        sk_hat, pk_hat = oracle.bfv_keygen_core(smp["ternary"], pk, smp["err"](), qs, psis, n)
        m = smp["rng"].integers(0, t, size=n, dtype=np.uint64)
        u = oracle.bfv_sample(qs, n, seed + 100)["ternary"]
        e = np.stack([smp["err"](), smp["err"]()])
        c = oracle.bfv_encrypt_core(np.stack([u, u]), pk_hat, e, m, qs, psis, n, t)
        got = oracle.bfv_decrypt(c.reshape(-1), sk_hat.reshape(-1)[: (R - 1) * n], qs, psis, n, t, gamma)
        assert np.array_equal(got, m)
