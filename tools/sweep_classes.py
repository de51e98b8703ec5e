#!/usr/bin/env python3
"""Throughput of every n = 2^15 kernel class (headroom class hl6 / hl4 / hl3 / hl2  x  near-2^k / general prime) at batch 1024.

A context's class is decided by its primes (kernels_fast.hip, fast_tables_create): hl = min over primes of
min(6, 64 - bit length) -> kernels <6>, <4> (hl 4, 5), <3> (hl 3, near-2^k primes only: round 4) or <2> (hl 2, and general primes at hl 3); "near" only if EVERY prime is 2^k - delta with
delta < 2^24.  Each class below is timed with four primes of that class (the reference's / BASELINE's where they exist,
otherwise the largest primes = 1 mod 2^16 below a bound far from a power of two, found here with Miller-Rabin; psi = the
minimal primitive 2^16-th root, as the reference's are).  Every set is checked on the spot: inverse(forward(a)) == a and
forward() of two sample polynomials against a Python evaluation of the transform at a few points.

usage: python tools/sweep_classes.py [num=1024]     (GPU)        -> profiles/r05_kernel_classes.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import params as P

N = 32768


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
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):          # deterministic below 3.3e24
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


def min_psi(q, n=N):
    """minimal primitive 2n-th root of unity mod q"""
    best = None
    for g in range(2, 2000):
        psi = pow(g, (q - 1) // (2 * n), q)
        if pow(psi, n, q) == q - 1:
            # all primitive 2n-th roots are odd powers of psi; the reference's are the numerically smallest
            best = psi
            break
    assert best is not None
    x, m = best, best
    sq = best * best % q
    for _ in range(n - 1):
        x = x * sq % q
        if x < m:
            m = x
    return m


def primes_below(bound, count):
    """the `count` largest primes = 1 mod 2^16 below bound"""
    out = []
    q = (bound >> 16 << 16) + 1
    while len(out) < count:
        q -= 1 << 16
        if is_prime(q):
            out.append(q)
    return out


def classes():
    c = {}
    c["hl6-near     (55-bit, demo.cu:35-36)"] = (P.Q55, P.PSI55)
    c["hl4-near     (60-bit, BASELINE)"] = (P.Q60, P.PSI60)
    # 58-bit: the reference's commented-out set (parameter.h:43-47, q = 2^58 - 2^24 + 2^18 + 1; its psi is for n = 4096, the
    # 2^16-th root is derived here) and three more primes just below 2^58: hl 6, near
    q58 = [P.REF_PARAMS_4096_58BIT[0]] + [q for q in primes_below(1 << 58, 4) if q != P.REF_PARAMS_4096_58BIT[0]][:3]
    c["hl6-near     (58-bit, parameter.h:43-47 + 3 more)"] = (q58, None)
    # general primes: far from a power of two (top bits 1011...), = 1 mod 2^16
    c["hl6-general  (57-bit)"] = (primes_below(0xB3 << 49, 4), None)
    c["hl6-general  (58-bit)"] = (primes_below(0xB3 << 50, 4), None)
    c["hl4-general  (60-bit)"] = (primes_below(0xB3 << 52, 4), None)
    c["hl5-near     (59-bit)"] = (primes_below(1 << 59, 4), None)
    c["hl3-near     (61-bit)"] = (primes_below(1 << 61, 4), None)
    c["hl3-general  (61-bit)"] = (primes_below(0xB3 << 53, 4), None)      # (class 2 until round 4)
    c["hl2-near     (62-bit)"] = (primes_below(1 << 62, 4), None)
    c["hl2-general  (62-bit)"] = (primes_below(0xB3 << 54, 4), None)
    return c


def main():
    import numpy as np
    import torch
    import ntt_cuda_amd as ntt

    num = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device("cuda", 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timeit(f, reps=40, warm=100):
        for _ in range(warm):
            f()
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    print("# n = 32768, %d polynomials per launch (%d MiB), context API, timed warm and back to back (40 launches after 100)" % (num, num * N * 8 >> 20))
    print("# %-52s %-6s %9s %9s %12s %14s %9s" % ("class (4 primes)", "kernel", "fwd ms", "inv ms", "pairs/s", "fused polymul/s", "TB/s inv"))
    rows = []
    for name, (qs, psis) in classes().items():
        if psis is None:
            psis = [min_psi(q) for q in qs]
        ctx = ntt.NTTContext(N, qs, psis)
        assert not ctx.uses_literal_kernels, name
        g = torch.Generator(device=dev).manual_seed(7)
        a = torch.empty((num, N), dtype=torch.int64, device=dev)
        for i, q in enumerate(qs):
            a[i::len(qs)] = torch.randint(0, q, a[i::len(qs)].shape, dtype=torch.int64, device=dev, generator=g) if q < (1 << 62) else \
                torch.randint(0, 1 << 62, a[i::len(qs)].shape, dtype=torch.int64, device=dev, generator=g) % q
        a0 = a.clone()
        b = a.flip(0).contiguous()
        # correctness on the spot: round trip, and NTT values of two polynomials at three points against direct evaluation
        ctx.forward_batch(a, num)
        A = ntt.to_host(a[:len(qs)].contiguous())
        h0 = ntt.to_host(a0[:len(qs)].contiguous())
        for i, (q, psi) in enumerate(zip(qs, psis)):
            for k in (0, 1, 12345):
                # output index k holds the evaluation at psi^(2*bitrev(k) + 1)
                br = int(format(k, "015b")[::-1], 2)
                x = pow(psi, 2 * br + 1, q)
                acc = 0
                for cf in reversed(h0[i].tolist()):
                    acc = (acc * x + int(cf)) % q
                assert int(A[i][k]) == acc, (name, i, k)
        ctx.inverse_batch(a, num)
        assert torch.equal(a, a0), name
        tf = timeit(lambda: ctx.forward_batch(a, num))
        ti = timeit(lambda: ctx.inverse_batch(a, num))

        def pair():
            ctx.forward_batch(a, num)
            ctx.inverse_batch(a, num)
        tp = timeit(pair)
        tm = timeit(lambda: ctx.polymul_batch(a, b, num), reps=20, warm=40)
        hl = min(min(6, 64 - q.bit_length()) for q in qs)
        kc = ctx.kernel_class                          # (what the library chose: mi355ntt_ctx_kernel_class)
        kern = "<%d,%s>" % (6 if kc[0] >= 6 else 5 if (kc[0] == 5 and kc[1]) else 4 if kc[0] >= 4 else 3 if kc[0] == 3 else 2, "near" if kc[1] else "gen")
        rows.append((name, tp))
        print("  %-52s %-6s %9.4f %9.4f %12.0f %14.0f %9.2f" % (name, kern, tf * 1e3, ti * 1e3, num / tp, num / tm, num * N * 16 / ti / 1e12))
        ctx.close()
    base = dict(rows)["hl4-near     (60-bit, BASELINE)"]
    print("# pairs/s relative to hl4-near: " + ", ".join("%s %.2f" % (n.split("(")[0].strip() + "(" + n.split("(")[1].split(",")[0].split(")")[0] + ")", base / t) for n, t in rows))


if __name__ == "__main__":
    main()
