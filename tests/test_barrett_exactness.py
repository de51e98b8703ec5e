"""Where "the same words as the reference" and "the exact transform" part ways.

The reference reduces every product with ONE conditional subtraction (singleBarrett, ntt_60bit.cuh:44-61).  For most
moduli that is exact; for a few (the second prime of its own decryption_test.cu among them) the quotient estimate can be
two short, the result comes out q too large, and the next butterfly's unsigned compare turns it into a wrong residue.
The oracle restates that arithmetic literally; the library predicts per modulus whether it can happen
(mi355ntt_barrett_is_exact) and, for such a modulus, runs the literal kernels unless asked for exact results."""
import numpy as np
import pytest

import params as P

KAT_Q = [68719403009, 68719230977, 137438822401]          # decryption_test.cu moduli
KAT_PSI = [24250113, 29008497, 8625844]
INEXACT = 68719230977                                      # 2^36 - 245759: frac(2^72 / q) = 0.88


def exact_forward(a, q, psi, n):
    """The transform the reference means to compute, in Python integers (same stage order and table as
    ntt_60bit.cuh:192-223, parameter.h:5-12, with every value fully reduced)."""
    lg = n.bit_length() - 1
    tab = [pow(psi, int(format(i, "0%db" % lg)[::-1], 2), q) for i in range(n)]
    a = [int(x) for x in a]
    length = 1
    while length < n:
        step = n // (2 * length)
        for p in range(length):
            w = tab[length + p]
            for j in range(2 * p * step, 2 * p * step + step):
                u, v = a[j], a[j + step] * w % q
                a[j], a[j + step] = (u + v) % q, (u - v) % q
        length *= 2
    return np.array(a, dtype=np.uint64)


def all_test_moduli():
    qs = [row[0] for row in P.REF_PARAMS.values()] + [P.REF_PARAMS_4096_58BIT[0], P.GAMMA61]
    qs += list(P.Q60) + list(P.Q55) + [v[0] for v in P.EDGE_PRIMES.values()] + [KAT_Q[0], KAT_Q[2]]
    return sorted(set(int(q) for q in qs))


def test_predicate_on_every_modulus_in_use(native):
    """Every modulus of BASELINE.json's configs, of parameter.h and of the edge-case tests is exact: for them the lazy
    kernels and the reference's arithmetic give the same words on all canonical inputs."""
    for q in all_test_moduli():
        assert native.barrett_is_exact(q), q
    assert not native.barrett_is_exact(INEXACT)
    assert not native.barrett_is_exact(1 << 62)                 # outside the supported range (q < 2^62)


def test_exact_moduli_never_under_reduce(native, oracle):
    """Adversarial operands (q-1, q-2, values near q) and random ones through the oracle's literal Barrett."""
    rng = np.random.default_rng(7)
    for q in all_test_moduli():
        k = oracle.lib().orc_bit_length(q)
        mu = oracle.lib().orc_mu(q, k)
        n = 1 << 16
        a = np.concatenate([np.full(n // 2, q - 1, dtype=np.uint64), q - 1 - rng.integers(0, 1 << 20, size=n // 2, dtype=np.uint64)])
        b = np.concatenate([rng.integers(0, q, size=n // 4, dtype=np.uint64), q - 1 - rng.integers(0, 1 << 20, size=n // 4, dtype=np.uint64)] * 2)
        c = np.empty_like(a)
        oracle.lib().orc_pointwise_batch(oracle._p(c), oracle._p(a), oracle._p(b), n, 1, 1, oracle._p(np.array([q], dtype=np.uint64)),
                                         oracle._p(np.array([mu], dtype=np.uint64)), oracle._p32(np.array([k], dtype=np.uint32)))
        want = np.array([int(x) * int(y) % q for x, y in zip(a[:2048], b[:2048])], dtype=np.uint64)
        assert c.max() < q and np.array_equal(c[:2048], want), q


def test_reference_arithmetic_under_reduces_on_the_inexact_modulus(oracle):
    """The witness: (q-1) * w through the literal Barrett returns a value >= q for ~1.8 % of w, and the forward transform
    of a ternary polynomial (entries 0, 1, q-1: what keygen_rns feeds it) then differs from the exact transform."""
    q, psi, n = INEXACT, KAT_PSI[1], 4096
    k = oracle.lib().orc_bit_length(q)
    mu = oracle.lib().orc_mu(q, k)
    rng = np.random.default_rng(3)
    w = rng.integers(0, q, size=4096, dtype=np.uint64)
    r = np.array([oracle.lib().orc_barrett(q - 1, int(x), q, mu, k) for x in w], dtype=np.uint64)
    over = r >= q
    assert 20 < over.sum() < 200                              # 1.8 % of 4096 = 74
    assert all((int(q - 1) * int(x)) % q == int(y) - q for x, y in zip(w[over], r[over]))     # exactly q too large
    prm = oracle.Params(n, [q], [psi])
    tern = oracle.bfv_sample([q], n, 1)["ternary"][0]
    lit = oracle.forward(tern, prm)
    exact = exact_forward(tern, q, psi, n)
    assert not np.array_equal(lit, exact)
    # on the exact neighbour modulus the same input agrees word for word
    prm0 = oracle.Params(n, [KAT_Q[0]], [KAT_PSI[0]])
    tern0 = oracle.bfv_sample([KAT_Q[0]], n, 1)["ternary"][0]
    assert np.array_equal(oracle.forward(tern0, prm0), exact_forward(tern0, KAT_Q[0], KAT_PSI[0], n))


@pytest.mark.gpu
def test_context_with_inexact_modulus_reproduces_the_reference_by_default(native, oracle, gpu):
    import torch
    n = 4096
    ctx = native.NTTContext(n, KAT_Q, KAT_PSI)
    assert ctx.uses_literal_kernels
    prm = oracle.Params(n, KAT_Q, KAT_PSI)
    tern = oracle.bfv_sample(KAT_Q, n, 1)["ternary"]
    want = oracle.forward_batch(tern, prm).reshape(3, n)
    assert (want[1] >= KAT_Q[1]).any()                       # the reference's non-canonical word is part of the expectation
    d = native.to_device(tern)
    ctx.forward_batch(d, 3)
    torch.cuda.synchronize()
    got = native.to_host(d).reshape(3, n)
    assert np.array_equal(got, want)
    ctx.inverse_batch(d, 3)
    assert np.array_equal(native.to_host(d).reshape(3, n), oracle.inverse_batch(want, prm).reshape(3, n))
    # fused product = the reference's own three-step sequence
    a = native.to_device(tern)
    b = oracle.synth_batch(n, 3, KAT_Q, 5)
    ctx.polymul_batch(a, native.to_device(b), 3)
    want_mul = oracle.inverse_batch(oracle.pointwise_batch(want, b, prm), prm)
    assert np.array_equal(native.to_host(a).reshape(-1), want_mul.reshape(-1))
    ctx.close()


@pytest.mark.gpu
def test_exact_flag_returns_the_exact_transform(native, oracle, gpu):
    import torch
    n = 4096
    ctx = native.NTTContext(n, KAT_Q, KAT_PSI, exact_on_inexact_primes=True)
    assert not ctx.uses_literal_kernels
    tern = oracle.bfv_sample(KAT_Q, n, 1)["ternary"]
    d = native.to_device(tern)
    ctx.forward_batch(d, 3)
    torch.cuda.synchronize()
    got = native.to_host(d).reshape(3, n)
    for i in range(3):
        assert np.array_equal(got[i], exact_forward(tern[i], KAT_Q[i], KAT_PSI[i], n))
    ctx.inverse_batch(d, 3)
    assert np.array_equal(native.to_host(d).reshape(3, n), tern)       # and the round trip is the identity
    ctx.close()


def test_predicate_is_sound_exhaustively_on_small_moduli(native):
    """Every odd modulus with 8..11 bits, EVERY pair of canonical operands through the reference's reduction (numpy
    restatement of singleBarrett): whenever the predicate says "exact" there is no under-reduction.  (The converse need
    not hold: the predicate is a sufficient condition.)"""
    flagged_inexact = truly_inexact = 0
    for k in range(8, 12):
        for q in range((1 << (k - 1)) + 1, 1 << k, 2):
            mu = (1 << (2 * k)) // q
            x = np.arange(q, dtype=np.uint64)
            a = np.outer(x, x).reshape(-1)
            s = ((a >> np.uint64(k - 2)) * np.uint64(mu)) >> np.uint64(k + 2)
            r = a - s * np.uint64(q)
            r = np.where(r >= q, r - np.uint64(q), r)
            bad = bool((r >= q).any())
            assert np.array_equal(np.where(r >= q, r - np.uint64(q), r), a % np.uint64(q))      # never off by more than one q
            exact = native.barrett_is_exact(q)
            assert not (exact and bad), q
            flagged_inexact += not exact
            truly_inexact += bad
    assert truly_inexact > 0 and flagged_inexact >= truly_inexact
