"""Round 5: batches of full rounds of the persistent grid plus a short tail at n = 2^15 -- the head runs the persistent kernels, the tail
the small-batch kernels (kernels_fast.hip, tail_split_head); the words must not depend on where a batch is cut."""
import os
import subprocess
import sys

import numpy as np
import pytest

import params as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sample(num, head_guess):
    ys = {0, 1, num // 2, num - 2, num - 1}
    for h in head_guess:
        ys |= {max(0, h - 1), min(num - 1, h), min(num - 1, h + 1)}
    return sorted(y for y in ys if 0 <= y < num)


@pytest.mark.parametrize("primes,num", [(4, 256 + 40), (4, 512 + 100), (3, 512 + 77), (5, 640), (5, 1024 + 33), (4, 256 + 96), (4, 256 + 97), (5, 512 + 98)])
def test_tail_split_forward_inverse_product_match_oracle(native, oracle, gpu, primes, num):
    """forward / inverse / fused product (one second operand per polynomial) on batches of k x 256 + r polynomials with 3, 4 and 5
    moduli (the head is cut back to a multiple of the prime count), both sides of the limit of the tail's length: sampled polynomials -- the
    first and last, and those around every possible cut -- against the oracle, the round trip over the whole batch."""
    import torch
    n = 32768
    qs = (P.Q60 + [P.Q60_SPECIAL])[:primes]
    psis = (P.PSI60 + [P.PSI60_SPECIAL])[:primes]
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    a = oracle.synth_batch(n, num, qs, 50 + num).reshape(num, n)
    b = oracle.synth_batch(n, num, qs, 90 + num).reshape(num, n)
    cuts = [(num // 256) * 256 - k for k in range(primes)]
    sample = _sample(num, cuts)
    d_a = native.to_device(a)
    ctx.forward_batch(d_a, num)
    A = native.to_host(d_a).reshape(num, n)
    for y in sample:
        assert np.array_equal(A[y], oracle.forward(a[y], prm, y % primes)), ("forward", y)
    d_b = native.to_device(b)
    ctx.inverse_batch(d_b, num)
    Bi = native.to_host(d_b).reshape(num, n)
    for y in sample:
        assert np.array_equal(Bi[y], oracle.inverse(b[y], prm, y % primes)), ("inverse", y)
    ctx.inverse_batch(d_a, num)
    assert np.array_equal(native.to_host(d_a).reshape(num, n), a)
    d_bh = native.to_device(b)
    ctx.forward_batch(d_bh, num)
    Bh = native.to_host(d_bh).reshape(num, n)
    d_f = native.to_device(a)
    ctx.polymul_batch(d_f, d_bh, num)
    F = native.to_host(d_f).reshape(num, n)
    for y in sample:
        one = oracle.Params(n, [qs[y % primes]], [psis[y % primes]], tables=False)
        want = oracle.inverse(oracle.pointwise_batch(A[y], Bh[y], one).reshape(-1), prm, y % primes)
        assert np.array_equal(F[y], want), ("polymul", y)
    ctx.close()


@pytest.mark.parametrize("num,group", [(590, 295), (590, 0), (540, 135), (1100, 550), (540, 15), (540, 40)])
def test_tail_split_shared_second_operands(native, oracle, gpu, num, group):
    """mi355ntt_polymul_batch_shared across the cut: the tail inside one key group (590 / 295, 540 / 135, 1100 / 550), a single
    group (group 0), a tail that starts on a group boundary and spans two groups (540 polynomials in groups of 15: cut at 510 = 34 x 15),
    and one the rule must leave whole (groups of 40: the tail would straddle a boundary it does not start on) -- every polynomial against
    the plain product with its own copy of the shared operand."""
    import torch
    n, qs, psis = 32768, P.Q60 + [P.Q60_SPECIAL], P.PSI60 + [P.PSI60_SPECIAL]
    R = 5
    ctx = native.NTTContext(n, qs, psis)
    groups = (num + group - 1) // group if group else 1
    a = oracle.synth_batch(n, num, qs, 7 + num).reshape(num, n)
    keys = oracle.synth_batch(n, groups * R, qs, 1234).reshape(groups * R, n)        # (any words below q: NTT-domain operands)
    d_keys = native.to_device(keys)
    d_a = native.to_device(a)
    ctx.polymul_batch_shared(d_a, d_keys, num, group=group)
    got = native.to_host(d_a).reshape(num, n)
    idx = [((y // group) * R if group else 0) + y % R for y in range(num)]
    full_b = native.to_device(keys[idx])
    d_ref = native.to_device(a)
    ctx.polymul_batch(d_ref, full_b, num)                                            # one operand per polynomial: the same words
    assert np.array_equal(got, native.to_host(d_ref).reshape(num, n))
    prm = oracle.Params(n, qs, psis)
    for y in (0, num // 2, num - 1):
        one = oracle.Params(n, [qs[y % R]], [psis[y % R]], tables=False)
        want = oracle.inverse(oracle.pointwise_batch(oracle.forward(a[y], prm, y % R), keys[idx[y]], one).reshape(-1), prm, y % R)
        assert np.array_equal(got[y], want), y
    ctx.close()


def test_tail_split_can_be_switched_off_and_gives_the_same_words(native, gpu):
    """MI355NTT_NO_TAIL_SPLIT=1 (A/B measurements) runs the whole batch on the persistent kernels: the same digest."""
    child = r'''
import os, sys, hashlib
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, ntt_cuda_amd as ntt, params as P
ctx = ntt.NTTContext(32768, P.Q60, P.PSI60)
a = torch.empty((612, 32768), dtype=torch.int64, device="cuda:0"); ctx.synth_splitmix(a, 612, 3)
b = a.flip(0).contiguous()
ctx.forward_batch(a, 612); f = a.clone(); ctx.forward_batch(b, 612); ctx.polymul_batch(a, b, 612); ctx.inverse_batch(b, 612)
torch.cuda.synchronize()
print("DIGEST", hashlib.sha256(f.cpu().numpy().tobytes() + a.cpu().numpy().tobytes() + b.cpu().numpy().tobytes()).hexdigest())
'''
    outs = []
    for off in (None, "1"):
        env = dict(os.environ)
        if off:
            env["MI355NTT_NO_TAIL_SPLIT"] = off
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + child], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert outs[0] == outs[1]


def test_tail_split_under_checked_raw_calls(native, oracle, gpu):
    """forwardNTT_batch / inverseNTT_batch (ntt_60bit.cuh:608,652) through the reference-signature entry points on a batch that
    is cut (256 + 44 polynomials): both launch sequences of a checked call carry the guard, so a table rewritten in place sends
    head AND tail to the literal kernels, and the restored table brings both back."""
    n, qs, psis, num = 32768, P.Q60, P.PSI60, 256 + 44
    threads = min(8, os.cpu_count() or 1)
    prm = oracle.Params(n, qs, psis)
    mod = native.Moduli(qs)
    dev = lambda x: native.to_device(np.ascontiguousarray(x))
    host = lambda t: native.to_host(t).reshape(num, n)
    native.raw_cache_clear()
    d_tp, d_ti = dev(prm.psi_tabs), dev(prm.psiinv_tabs)
    assert native.raw_uses_fast_kernels(n, d_tp, mod) and native.raw_uses_fast_kernels(n, d_ti, mod, inverse=True)
    a = oracle.synth_batch(n, num, qs, 505).reshape(num, n)
    want = oracle.forward_batch(a, prm, threads=threads).reshape(num, n)
    d_a = dev(a)
    native.forwardNTT_batch(d_a, n, d_tp, num, 4, mod)
    assert np.array_equal(host(d_a), want)
    native.inverseNTT_batch(d_a, n, d_ti, num, 4, mod)
    assert np.array_equal(host(d_a), a)
    tabs = prm.psi_tabs.copy()
    tabs[1, 20000] = (int(tabs[1, 20000]) + 5) % qs[1]      # (an entry of the last stage: one butterfly per polynomial of prime 1 changes)
    d_tp.copy_(dev(tabs).reshape(d_tp.shape))
    prm2 = oracle.Params(n, qs, psis)
    prm2.psi_tabs[:] = tabs
    want2 = oracle.forward_batch(a, prm2, threads=threads).reshape(num, n)
    assert not np.array_equal(want2[257], want[257])      # (a polynomial of the tail is affected)
    d_a = dev(a)
    native.forwardNTT_batch(d_a, n, d_tp, num, 4, mod)
    assert np.array_equal(host(d_a), want2)
    d_tp.copy_(dev(prm.psi_tabs).reshape(d_tp.shape))
    d_a = dev(a)
    native.forwardNTT_batch(d_a, n, d_tp, num, 4, mod)
    assert np.array_equal(host(d_a), want)


def test_tail_split_in_a_captured_graph(native, oracle, gpu):
    """head + tail launch sequences captured into one hipGraph (forward, fused product, inverse on 512 + 60 polynomials) and replayed
    twice on fresh data: the oracle's words every time."""
    import torch
    n, qs, psis, num = 32768, P.Q60, P.PSI60, 512 + 60
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    a = oracle.synth_batch(n, num, qs, 901).reshape(num, n)
    b = oracle.synth_batch(n, num, qs, 902).reshape(num, n)
    d_a, d_b, d_c = native.to_device(a), native.to_device(b), native.to_device(a)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.Stream()):
        ctx.forward_batch(d_b, num)                    # b -> bhat
        ctx.polymul_batch(d_a, d_b, num)               # a <- INTT(NTT(a) . bhat)
        ctx.forward_batch(d_c, num)
        ctx.inverse_batch(d_c, num)                    # round trip
    sample = _sample(num, [512])
    for rep in range(2):
        d_a.copy_(native.to_device(a)); d_b.copy_(native.to_device(b)); d_c.copy_(native.to_device(a))
        g.replay()
        torch.cuda.synchronize()
        A, Bh, C = (native.to_host(t).reshape(num, n) for t in (d_a, d_b, d_c))
        assert np.array_equal(C, a), rep
        for y in sample:
            bh = oracle.forward(b[y], prm, y % 4)
            assert np.array_equal(Bh[y], bh), (rep, y)
            prod = oracle.pointwise_batch(oracle.forward(a[y], prm, y % 4), bh, oracle.Params(n, [qs[y % 4]], [psis[y % 4]], tables=False)).reshape(-1)
            assert np.array_equal(A[y], oracle.inverse(prod, prm, y % 4)), (rep, y)
    ctx.close()


@pytest.mark.parametrize("num", [4095, 4096, 4097])
def test_inverse_on_both_sides_of_the_streaming_switch(native, oracle, gpu, num):
    """k_inverse15 reads its rows with the non-temporal hint from 4096 polynomials up (kStreamLoads, kernels.hpp): the words must not
    depend on the policy -- round trip over the whole batch on the device, sampled polynomials against the oracle."""
    import torch
    n, qs, psis = 32768, P.Q60, P.PSI60
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    a = torch.empty((num, n), dtype=torch.int64, device="cuda:0")
    ctx.synth_splitmix(a, num, 1)
    a0 = a.clone()
    ctx.inverse_batch(a, num)                     # (any canonical words are a valid input)
    sample = _sample(num, [4096, 2048])
    got = {y: a[y].cpu().numpy().view(np.uint64) for y in sample}
    for y in sample:
        assert np.array_equal(got[y], oracle.inverse(a0[y].cpu().numpy().view(np.uint64), prm, y % 4)), (num, y)
    ctx.forward_batch(a, num)
    assert torch.equal(a, a0), num
    ctx.close()


@pytest.mark.parametrize("num", [512, 513, 700])
def test_fused_product_on_both_sides_of_the_streaming_switch(native, oracle, gpu, num):
    """k_polymul15 reads own second operands with the non-temporal hint above 512 polynomials; shared ones never (kSharedB)."""
    import torch
    n, qs, psis = 32768, P.Q60, P.PSI60
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    a = torch.empty((num, n), dtype=torch.int64, device="cuda:0")
    b = torch.empty((num, n), dtype=torch.int64, device="cuda:0")
    ctx.synth_splitmix(a, num, 11)
    ctx.synth_splitmix(b, num, 50011)
    a0 = a.clone()
    ctx.forward_batch(b, num)                     # bhat
    ctx.polymul_batch(a, b, num)
    # the same product composed of the three steps
    c = a0.clone()
    ctx.forward_batch(c, num)
    ctx.pointwise_mul(c, c, b, num)
    ctx.inverse_batch(c, num)
    assert torch.equal(a, c), num
    for y in _sample(num, [512]):
        ah = oracle.forward(a0[y].cpu().numpy().view(np.uint64), prm, y % 4)
        prod = oracle.pointwise_batch(ah, b[y].cpu().numpy().view(np.uint64), oracle.Params(n, [qs[y % 4]], [psis[y % 4]], tables=False)).reshape(-1)
        assert np.array_equal(a[y].cpu().numpy().view(np.uint64), oracle.inverse(prod, prm, y % 4)), (num, y)
    ctx.close()
