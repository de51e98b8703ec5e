"""Round 6: kernel class 0 -- the reference's own arithmetic (singleBarrett with one conditional subtraction, the value carried from
stage to stage as the reference's memory carries it, a halving in every inverse stage; ntt_60bit.cuh:44-61,199-222,232-264) in the
single-pass, register-resident kernel shape (csrc/kernels_lit.cuh): contexts with a Barrett-inexact prime, every ring degree of the
reference's dispatch, word for word against the oracle."""
import numpy as np
import pytest

import params as P

pytestmark = pytest.mark.gpu


def _adversarial(a, qs, rng):
    """rows of {0, 1, q - 1, q - 2} (what ternary keys feed the transforms: the operand q - 1 is where the reference's Barrett
    under-reduces most often) mixed into uniform residues"""
    num, n = a.shape
    for y in range(0, num, 3):
        q = int(qs[y % len(qs)])
        pick = rng.integers(0, 4, size=n)
        a[y] = np.array([0, 1, q - 1, q - 2], dtype=np.uint64)[pick]
    return a


def _root(q, n):
    """a primitive 2n-th root of unity mod q (q = 1 mod 2n)"""
    assert (q - 1) % (2 * n) == 0
    for g in range(2, 1000):
        w = pow(g, (q - 1) // (2 * n), q)
        if pow(w, n, q) == q - 1:
            return w
    raise ValueError(q)


def _moduli(kind, n):
    if kind == "inexact60":            # one Barrett-inexact 60-bit modulus: every polynomial through the literal butterflies
        q, r = P.INEXACT_PRIMES[60]
        return [q], [r[n]]
    if kind == "mixed":                # inexact 36-bit, exact 60-bit, inexact 61-bit, exact 36-bit: both passes of a workgroup's walk
        sel = [P.INEXACT_PRIMES[36], P.EXACT_NEIGHBOURS[60], P.INEXACT_PRIMES[61], P.EXACT_NEIGHBOURS[36]]
        if n == 65536:                 # (the exact neighbours are 1 mod 2^16 only: the bench's 60-bit primes stand in)
            sel = [P.INEXACT_PRIMES[36], (P.Q60[0], {n: _root(P.Q60[0], n)}), P.INEXACT_PRIMES[61], (P.Q60[1], {n: _root(P.Q60[1], n)})]
        return [q for q, _ in sel], [r[n] for _, r in sel]
    if kind == "inexact3":             # three inexact moduli of 34, 50 and 60 bits
        sel = [P.INEXACT_PRIMES[34], P.INEXACT_PRIMES[50], P.INEXACT_PRIMES[60]]
        return [q for q, _ in sel], [r[n] for _, r in sel]
    raise KeyError(kind)


@pytest.mark.parametrize("n", [2048, 4096, 8192, 16384, 32768, 65536])
@pytest.mark.parametrize("kind", ["inexact60", "mixed", "inexact3"])
def test_literal_class_matches_oracle_word_for_word(native, oracle, gpu, n, kind):
    """forward, inverse and the fused product of a class-0 context: every word the oracle's, non-canonical ones included, for a few
    polynomials and for a batch that makes the persistent workgroups walk more than one polynomial of each kind."""
    import torch
    qs, psis = _moduli(kind, n)
    assert any(not native.barrett_is_exact(q) for q in qs)
    ctx = native.NTTContext(n, qs, psis)
    assert ctx.kernel_class == (0, False) or ctx.kernel_class == 0, ctx.kernel_class
    assert ctx.literal_routing == (2 if kind == "mixed" else 1)
    prm = oracle.Params(n, qs, psis)
    rng = np.random.default_rng(n + len(qs))
    per_cu = {2048: 16, 4096: 8, 8192: 4, 16384: 2, 32768: 1, 65536: 1}[n]
    big = 256 * per_cu * 2 + 5 * len(qs) + 1 if n <= 4096 else (256 * per_cu + 67 if n <= 32768 else 131)      # beyond one polynomial per resident workgroup
    seen_noncanonical = False
    # (n = 2^15: two whole rounds of the persistent grid -- 255 workgroups when the prime count shares a factor with 256 -- take the
    # single-pass shape by the default rule, the other sizes the small-batch shape)
    for num in (1, len(qs) + 2, big) + (((510 if kind == "mixed" else 512),) if n == 32768 else ()):
        a = _adversarial(oracle.synth_batch(n, num, qs, 300 + num).reshape(num, n), qs, rng)
        want_f = oracle.forward_batch(a.copy(), prm, threads=8).reshape(num, n)
        d = native.to_device(a)
        ctx.forward_batch(d, num)
        torch.cuda.synchronize()
        got = native.to_host(d).reshape(num, n)
        bad = np.nonzero((got != want_f).any(axis=1))[0]
        assert bad.size == 0, ("forward", n, kind, num, bad[:8])
        qcol = np.array(qs, dtype=np.uint64)[np.arange(num) % len(qs)][:, None]
        seen_noncanonical |= bool((want_f >= qcol).any())
        ctx.inverse_batch(d, num)
        want_i = oracle.inverse_batch(want_f.copy(), prm, threads=8).reshape(num, n)
        got = native.to_host(d).reshape(num, n)
        bad = np.nonzero((got != want_i).any(axis=1))[0]
        assert bad.size == 0, ("inverse", n, kind, num, bad[:8])
        if num != big or n <= 8192 or n == 65536:
            b = oracle.synth_batch(n, num, qs, 700 + num).reshape(num, n)
            da = native.to_device(a)
            ctx.polymul_batch(da, native.to_device(b), num)
            want_m = oracle.inverse_batch(oracle.pointwise_batch(want_f, b, prm), prm, threads=8).reshape(num, n)
            got = native.to_host(da).reshape(num, n)
            bad = np.nonzero((got != want_m).any(axis=1))[0]
            assert bad.size == 0, ("polymul", n, kind, num, bad[:8])
    if kind == "inexact60":     # (how often the reference under-reduces depends on frac(2^(2k) / q): this modulus does on every batch)
        assert seen_noncanonical, "the reference's non-canonical words must be part of the expectation"
    ctx.close()


@pytest.mark.parametrize("n", [4096, 32768])
def test_literal_class_follows_the_reference_on_arbitrary_words(native, oracle, gpu, n):
    """A context whose primes are ALL inexact runs the literal butterflies on every polynomial: the same 64-bit operations on the
    same operands as the reference's stage kernels, so even words that are no residues at all (q + r, 2^64 - 1, random 64-bit
    patterns) come out as the oracle's."""
    import torch
    qs, psis = _moduli("inexact3", n)
    ctx = native.NTTContext(n, qs, psis)
    prm = oracle.Params(n, qs, psis)
    rng = np.random.default_rng(5)
    num = 7
    a = rng.integers(0, 1 << 64, size=(num, n), dtype=np.uint64)
    a[1] = np.uint64((1 << 64) - 1)
    a[2] = np.array([int(qs[2]) + int(x) for x in rng.integers(0, 1000, size=n)], dtype=np.uint64)
    for op, fn in (("forward", oracle.forward_batch), ("inverse", oracle.inverse_batch)):
        d = native.to_device(a)
        getattr(ctx, op + "_batch")(d, num)
        torch.cuda.synchronize()
        assert np.array_equal(native.to_host(d).reshape(num, n), fn(a.copy(), prm).reshape(num, n)), op
    ctx.close()


def test_raw_calls_on_an_inexact_modulus_run_the_single_pass_kernels(native, oracle, gpu):
    """forwardNTT_batch / inverseNTT_batch with the caller's tables (ntt_60bit.cuh:608,652) on a set with a Barrett-inexact modulus:
    the derived context is a class-0 one -- checked, trusted and captured calls all return the oracle's words; a table rewritten in
    place sends the call to the guarded fallback leg."""
    import torch
    n = 32768
    qs, psis = _moduli("mixed", n)
    prm = oracle.Params(n, qs, psis)
    mod = native.Moduli(qs)
    tf = torch.from_numpy(prm.psi_tabs.view(np.int64).reshape(len(qs), n)).cuda()
    ti = torch.from_numpy(prm.psiinv_tabs.view(np.int64).reshape(len(qs), n)).cuda()
    num = 41
    a = _adversarial(oracle.synth_batch(n, num, qs, 9).reshape(num, n), qs, np.random.default_rng(1))
    want_f = oracle.forward_batch(a.copy(), prm, threads=8).reshape(num, n)
    want_i = oracle.inverse_batch(want_f.copy(), prm, threads=8).reshape(num, n)
    d = native.to_device(a)
    native.forwardNTT_batch(d, n, tf, num, len(qs), mod)
    assert native.raw_uses_fast_kernels(n, tf, mod)
    assert np.array_equal(native.to_host(d).reshape(num, n), want_f)
    native.inverseNTT_batch(d, n, ti, num, len(qs), mod)
    assert np.array_equal(native.to_host(d).reshape(num, n), want_i)
    # a table rewritten in place (another root): the checked call must follow the caller's table, not the cached context
    psis2 = [pow(int(w), 3, int(q)) for w, q in zip(psis, qs)]
    prm2 = oracle.Params(n, qs, psis2)
    tf.copy_(torch.from_numpy(prm2.psi_tabs.view(np.int64).reshape(len(qs), n)))
    d = native.to_device(a)
    native.forwardNTT_batch(d, n, tf, num, len(qs), mod)
    assert np.array_equal(native.to_host(d).reshape(num, n), oracle.forward_batch(a.copy(), prm2, threads=8).reshape(num, n))
    native.raw_cache_clear()
    # captured and replayed: checked calls inside a capture follow the caller's table with the literal stage kernels, trusted ones run
    # the single-pass kernels (nothing is shared between the graph and other streams then) -- the same words both ways
    tf.copy_(torch.from_numpy(prm.psi_tabs.view(np.int64).reshape(len(qs), n)))
    d = native.to_device(a)
    src = native.to_device(a)
    native.forwardNTT_batch(d, n, tf, num, len(qs), mod)          # first sight outside the capture (derives the contexts)
    native.inverseNTT_batch(d, n, ti, num, len(qs), mod)
    torch.cuda.synchronize()
    g, cs = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.graph(g, stream=cs):
        native.forwardNTT_batch(d, n, tf, num, len(qs), mod)
        native.inverseNTT_batch(d, n, ti, num, len(qs), mod)
    for _ in range(2):
        d.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(native.to_host(d).reshape(num, n), want_i)
    assert native.raw_trust_tables(n, tf, mod) and native.raw_trust_tables(n, ti, mod, inverse=True)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=cs):
        native.forwardNTT_batch(d, n, tf, num, len(qs), mod)
        native.inverseNTT_batch(d, n, ti, num, len(qs), mod)
    for _ in range(2):
        d.copy_(src)
        g2.replay()
        torch.cuda.synchronize()
        assert np.array_equal(native.to_host(d).reshape(num, n), want_i)
    native.raw_cache_clear()


def test_checked_raw_calls_share_nothing_between_streams(native, oracle, gpu):
    """Checked forwardNTT_batch / inverseNTT_batch calls with ONE table from twenty streams -- more than an entry keeps guard records
    for (16: the others run the literal stage kernels), interleaved so that every call finds the entry last used by another stream,
    some of the streams gone by the time the next call arrives: every word the oracle's, also after the table was rewritten in place
    (the comparison's words are per stream: no call waits for, or records on, a stream that is not its own)."""
    import gc
    import torch
    n, num = 4096, 24
    qs = list(P.Q60)
    psis = [pow(w, 32768 // n, q) for w, q in zip(P.PSI60, qs)]          # (PSI60: primitive 2 * 32768-th roots)
    prm = oracle.Params(n, qs, psis)
    mod = native.Moduli(qs)
    tf = torch.from_numpy(prm.psi_tabs.view(np.int64).reshape(len(qs), n)).cuda()
    ti = torch.from_numpy(prm.psiinv_tabs.view(np.int64).reshape(len(qs), n)).cuda()
    a = oracle.synth_batch(n, num, qs, 77).reshape(num, n)
    want_f = oracle.forward_batch(a.copy(), prm).reshape(num, n)
    streams = [torch.cuda.Stream() for _ in range(20)]
    bufs = [native.to_device(a) for _ in streams]
    torch.cuda.synchronize()
    for rep in range(3):
        for s, d in zip(streams, bufs):
            native.forwardNTT_batch(d, n, tf, num, len(qs), mod, stream=s)
        for i, (s, d) in enumerate(zip(streams, bufs)):
            s.synchronize()
            assert np.array_equal(native.to_host(d).reshape(num, n), want_f), ("forward", rep, i)
        for s, d in zip(reversed(streams), reversed(bufs)):
            native.inverseNTT_batch(d, n, ti, num, len(qs), mod, stream=s)
        for i, (s, d) in enumerate(zip(streams, bufs)):
            s.synchronize()
            assert np.array_equal(native.to_host(d).reshape(num, n), a), ("inverse", rep, i)
        if rep == 0:                    # half of the streams go away; new ones (possibly with the same handles) take their place
            for k in range(0, 20, 2):
                streams[k] = None
            gc.collect()
            for k in range(0, 20, 2):
                streams[k] = torch.cuda.Stream()
    # the table rewritten in place: every stream's next call follows it
    psis2 = [pow(int(w), 5, int(q)) for w, q in zip(psis, qs)]
    prm2 = oracle.Params(n, qs, psis2)
    torch.cuda.synchronize()
    tf.copy_(torch.from_numpy(prm2.psi_tabs.view(np.int64).reshape(len(qs), n)))
    torch.cuda.synchronize()
    want2 = oracle.forward_batch(a.copy(), prm2).reshape(num, n)
    for s, d in zip(streams, bufs):
        native.forwardNTT_batch(d, n, tf, num, len(qs), mod, stream=s)
    for i, (s, d) in enumerate(zip(streams, bufs)):
        s.synchronize()
        assert np.array_equal(native.to_host(d).reshape(num, n), want2), ("rewritten table", i)
    native.raw_cache_clear()


@pytest.mark.parametrize("forced", ["0", "1000000"])
def test_both_kernel_shapes_of_class_0_at_every_batch_size(forced):
    """Class 0 has two kernel shapes -- single-pass (one workgroup per polynomial) and small-batch (a polynomial over n/512 waves, two
    launches per transform) -- and picks by batch size.  MI355NTT_LATENCY_PATH_MAX (read once per process) forces one of them for every
    size: the word-for-word tests above, the moduli sweep and the KAT-1 routing test pass either way."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI355NTT_LATENCY_PATH_MAX=forced)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_round6.py"), os.path.join(root, "tests", "test_gpu_round4.py"),
                        "-k", "word_for_word or arbitrary_words or per_prime_literal_routing"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
