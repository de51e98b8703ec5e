"""Randomised differential sweep over moduli (round 5).

The kernel a context runs is decided per modulus at run time: the headroom class from the bit length, "near 2^k" from three
inequalities on d = 2^k - q (kernels_fast.hip, fast_tables_create: d < 2^24, 2^(64-k) d + 2 d < 2^k, 2 d^2 + 3 d < 2^k), and
throughput or literal kernels from the exactness of the reference's single-subtraction Barrett (hostparams.cpp,
barrett_single_subtraction_exact).  The hand-picked moduli of test_gpu_parity.py sit well inside their classes; here the moduli
are DRAWN -- per bit length 50..62 a few primes = 1 (mod 2^17) from a fixed seed -- and CONSTRUCTED on both sides of every
threshold (the nearest primes below and above it), the class the library took is asserted against an independent computation,
and forward / inverse / fused product run against the oracle at n = 2^11, 2^13, 2^15, 2^16 on both sides of the small-batch
switch, with the adversarial coefficient pattern of test_gpu_parity.py."""
import random
from fractions import Fraction

import numpy as np
import pytest

STEP = 1 << 17          # q = 1 (mod 2^17): a 2n-th root of unity exists up to n = 2^16


def is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):      # deterministic below 3.3e24
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def psi_for(q, n):
    for x in range(2, 2000):
        psi = pow(x, (q - 1) // (2 * n), q)
        if pow(psi, n, q) == q - 1:
            return psi
    raise AssertionError("no 2n-th root found")


def near_ok(q):
    """fast_tables_create's test, recomputed with Python integers"""
    k = q.bit_length()
    d = (1 << k) - q
    return k > 32 and d < (1 << 24) and (d << (64 - k)) + 2 * d < (1 << k) and 2 * d * d + 3 * d < (1 << k)


def barrett_margin(q):
    """bound of barrett_single_subtraction_exact as an exact fraction: the single subtraction is exact when it is below 1"""
    k = q.bit_length()
    mu = (1 << (2 * k)) // q
    f = Fraction((1 << (2 * k)) % q, q)
    top = Fraction(q - 1, 1 << k)
    return top * top * f + Fraction(mu, 1 << (k + 2))


def expected_class(qs):
    return min(min(64 - q.bit_length() for q in qs), 6), all(near_ok(q) for q in qs)


def drawn_primes():
    """per bit length 50..62: two primes = 1 (mod 2^17) drawn uniformly from the upper part of the range (fixed seed)"""
    rng = random.Random(20261003)
    out = []
    for k in range(50, 63):
        got = 0
        while got < 2:
            m = rng.randrange((1 << (k - 1)) // STEP + 1, (1 << k) // STEP)
            q = m * STEP + 1
            if q.bit_length() == k and is_prime(q) and abs(barrett_margin(q) - 1) > Fraction(1, 10 ** 6):
                out.append(("drawn-%d-%d" % (k, got), q))
                got += 1
    return out


def threshold_primes():
    """the nearest primes on both sides of each run-time threshold"""
    out = []
    # (1) near-2^k: d = 2^k - q = 2^17 j - 1.  For k >= 50 only d < 2^24 binds (j <= 128); for k = 41 .. 47 one of the two other
    # inequalities flips first.  Walk j upwards and keep the last prime that qualifies and the first that does not.
    for k in (41, 42, 43, 45, 47, 50, 55, 59, 60, 61, 62):
        last_in = first_out = None
        for j in range(1, 4000):
            q = (1 << k) - (STEP * j - 1)
            if q.bit_length() != k or not is_prime(q):
                continue
            if near_ok(q):
                if first_out is None:
                    last_in = q
            elif first_out is None:
                first_out = q
            if first_out is not None and last_in is not None:
                break
        if last_in:
            out.append(("near-in-%d" % k, last_in))
        if first_out:
            out.append(("near-out-%d" % k, first_out))
    # (1b) the inequality 2^(64-k) d + 2 d < 2^k flips before the two others only for k <= 42, where primes = 1 (mod 2^17) leave it
    # a handful of candidates: primes = 1 (mod 2^12) instead (n = 2048 only)
    for k in (40, 41, 42):
        last_in = first_out = None
        for j in range(1, 4000):
            q = (1 << k) - ((1 << 12) * j - 1)
            d = (1 << k) - q
            if q.bit_length() != k or not is_prime(q):
                continue
            if near_ok(q):
                if first_out is None:
                    last_in = q
            elif first_out is None and d < (1 << 24) and 2 * d * d + 3 * d < (1 << k):     # (only the middle inequality fails)
                first_out = q
            if first_out is not None and last_in is not None:
                break
        assert last_in and first_out, k
        out += [("near-in-c-%d" % k, last_in), ("near-out-c-%d" % k, first_out)]
    # (2) Barrett exactness: primes whose bound lies within 2 % of 1, two on either side (drawn from the top of the 55- to 61-bit
    # ranges, where about one prime in five is inexact)
    rng = random.Random(7)
    lo, hi = [], []
    while len(lo) < 2 or len(hi) < 2:
        k = rng.choice((55, 58, 60, 61))
        m = rng.randrange(int((1 << k) * 0.9) // STEP, (1 << k) // STEP)
        q = m * STEP + 1
        if q.bit_length() != k or not is_prime(q):
            continue
        b = barrett_margin(q)
        if Fraction(98, 100) < b < Fraction(999999, 1000000) and len(lo) < 2:
            lo.append(q)
        elif Fraction(1000001, 1000000) < b < Fraction(102, 100) and len(hi) < 2:
            hi.append(q)
    out += [("barrett-exact-by-2pct-%d" % i, q) for i, q in enumerate(lo)] + [("barrett-inexact-by-2pct-%d" % i, q) for i, q in enumerate(hi)]
    return out


CASES = drawn_primes() + threshold_primes()
# both sides of the small-batch switch (kernels_fast_impl.cuh, use_latency_path; n = 2^16: forward pair from 72, product from 96,
# inverse from 120 polynomials)
BATCHES = {2048: (3, 520), 8192: (3, 300), 32768: (3, 200), 65536: (3, 130)}


def test_the_sweep_covers_every_class_and_both_sides_of_every_threshold():
    classes = {expected_class([q]) for _, q in CASES}
    assert {(6, True), (6, False), (5, True), (4, True), (4, False), (3, True), (2, True), (2, False)} <= classes, classes
    names = [nm for nm, _ in CASES]
    for k in (45, 47, 50, 55, 59, 60, 61, 62):            # d < 2^24 (k >= 50) and 2 d^2 + 3 d < 2^k (k = 45, 47)
        assert "near-in-%d" % k in names and "near-out-%d" % k in names, k
    for k in (40, 41, 42):                                # 2^(64-k) d + 2 d < 2^k
        assert "near-in-c-%d" % k in names and "near-out-c-%d" % k in names, k
    assert sum(nm.startswith("barrett-inexact") for nm in names) == 2 and sum(nm.startswith("barrett-exact") for nm in names) == 2
    assert all(q % (1 << 12) == 1 and is_prime(q) for _, q in CASES)


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(CASES)), ids=[nm for nm, _ in CASES])
def test_modulus_against_oracle(native, oracle, gpu, idx):
    name, q = CASES[idx]
    exact = barrett_margin(q) < 1
    assert bool(native.barrett_is_exact(q)) == exact, (name, q)
    sizes = [2048, 8192, 32768, 65536] if q % STEP == 1 else [2048]
    for si, n in enumerate(sizes):
        psi = psi_for(q, n)
        prm = oracle.Params(n, [q], [psi])
        ctx = native.NTTContext(n, [q], [psi])
        # (Barrett-inexact moduli of 34 ... 61 bits run the reference's own arithmetic in the single-pass kernels up to n = 2^15: class 0)
        single_pass_literal = (not exact) and 34 <= q.bit_length() <= 61      # (n = 2^16: two half-size class-0 transforms around the literal coupling stage)
        assert ctx.kernel_class == ((0, False) if single_pass_literal else expected_class([q])), (name, q, n, ctx.kernel_class)
        assert ctx.literal_routing == (0 if exact else 1), (name, q, n)
        small, large = BATCHES[n]
        # every modulus runs the small batch at every size and the large batch at two of the four sizes (by turns)
        for num in (small, large) if (idx + si) % 2 == 0 else (small,):
            a = oracle.synth_batch(n, num, [q], 1000 + idx).reshape(num, n)
            b = oracle.synth_batch(n, num, [q], 2000 + idx).reshape(num, n)
            for y in range(min(num, 2)):
                for arr in (a, b):
                    arr[y, :8] = [0, 1, q - 1, q - 2, q - 1, 0, q - 1, 1]
                    arr[y, n // 2 - 2: n // 2 + 2] = [q - 1, 0, q - 1, q - 1]
                    arr[y, n - 4:] = [q - 1, q - 1, 0, q - 2]
                    arr[y, 1000:1000 + 64] = q - 1
            sample = sorted({0, 1, num // 2, num - 1})
            d_a, d_b = native.to_device(a), native.to_device(b)
            ctx.forward_batch(d_a, num)
            A = native.to_host(d_a).reshape(num, n)
            for y in sample:
                assert np.array_equal(A[y], oracle.forward(a[y], prm, 0)), (name, q, n, num, "forward", y)
            ctx.inverse_batch(d_b, num)                  # (any words below q are a valid input)
            Bi = native.to_host(d_b).reshape(num, n)
            for y in sample:
                assert np.array_equal(Bi[y], oracle.inverse(b[y], prm, 0)), (name, q, n, num, "inverse", y)
            ctx.inverse_batch(d_a, num)
            back = native.to_host(d_a).reshape(num, n)
            if exact:
                assert np.array_equal(back, a), (name, q, n, num, "round trip")
            else:                                        # (the reference's own words: a transform pair need not be the identity there)
                for y in sample:
                    assert np.array_equal(back[y], oracle.inverse(A[y], prm, 0)), (name, q, n, num, "inverse of forward", y)
            d_bh = native.to_device(b)
            ctx.forward_batch(d_bh, num)
            Bh = native.to_host(d_bh).reshape(num, n)
            d_f = native.to_device(a)
            ctx.polymul_batch(d_f, d_bh, num)
            F = native.to_host(d_f).reshape(num, n)
            one = oracle.Params(n, [q], [psi], tables=False)
            for y in sample:
                want = oracle.inverse(oracle.pointwise_batch(oracle.forward(a[y], prm, 0), Bh[y], one).reshape(-1), prm, 0)
                assert np.array_equal(F[y], want), (name, q, n, num, "polymul", y)
        ctx.close()


@pytest.mark.gpu
def test_mixed_classes_take_the_weakest(native, oracle, gpu):
    """a context is instantiated for the minimum headroom over its primes and is near-2^k only if every prime is"""
    by = {}
    for nm, q in CASES:
        if barrett_margin(q) < 1:
            by.setdefault(expected_class([q]), q)
    n = 8192
    for combo in ([(6, True), (4, True)], [(6, True), (6, False)], [(4, True), (2, False), (6, True)], [(5, True), (3, True)]):
        qs = [by[c] for c in combo]
        psis = [psi_for(q, n) for q in qs]
        ctx = native.NTTContext(n, qs, psis)
        assert ctx.kernel_class == expected_class(qs), (combo, ctx.kernel_class)
        prm = oracle.Params(n, qs, psis)
        num = 3 * len(qs) + 1
        a = oracle.synth_batch(n, num, qs, 5)
        d_a = native.to_device(a)
        ctx.forward_batch(d_a, num)
        assert np.array_equal(native.to_host(d_a), oracle.forward_batch(a, prm)), combo
        ctx.inverse_batch(d_a, num)
        assert np.array_equal(native.to_host(d_a), a), combo
        ctx.close()
