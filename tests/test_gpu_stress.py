"""GPU soak (-m gpu, well under a minute on an MI355X): the concurrent code paths run hundreds of times over random sizes, batch counts and streams.

Round 2's advisor found an LDS race in k_inverse<LOGN> that 126 single-shot GPU tests had not exposed; round 3 added cross-workgroup
flags, counted vmcnt waits around LDS-direct loads and a watchdog.  Each configuration's expected words are pinned ONCE -- a sample of
polynomials against the CPU oracle, the rest by the first GPU result -- and every later iteration must reproduce them bit for bit:
a race shows as a mismatch, a lost wake-up as a hang / watchdog abort.  One variant takes half of the CUs away with a foreign kernel on
another stream while the cooperating-workgroup launches run (their grid is then NOT resident as a whole): slow, not wrong, no trap.
"""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import params as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _psi_for(q, n):
    return next(x for x in (pow(g, (q - 1) // (2 * n), q) for g in range(2, 2000)) if pow(x, n, q) == q - 1)


def _dev(torch, gpu, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(gpu)


class _Case:
    """inputs of `cap` polynomials and their forward transforms on the device; the oracle pins a sample"""

    def __init__(self, torch, native, oracle, gpu, n, qs, cap, seed, sample=3):
        self.n, self.qs, self.cap, self.P = n, qs, cap, len(qs)
        psis = [_psi_for(q, n) for q in qs]
        self.ctx = native.NTTContext(n, qs, psis)
        self.a = torch.empty((cap, n), dtype=torch.int64, device=gpu)
        self.ctx.synth_splitmix(self.a, cap, seed)
        self.A = self.a.clone()
        self.ctx.forward_batch(self.A, cap)
        torch.cuda.synchronize()
        prm = oracle.Params(n, qs, psis)
        host_a = oracle.synth_batch(n, cap, qs, seed).reshape(cap, n)
        for y in sorted({0, cap // 2 + 1, cap - 1})[:sample]:
            assert np.array_equal(native.to_host(self.a[y].contiguous()), host_a[y]), ("synth", n, y)
            assert np.array_equal(native.to_host(self.A[y].contiguous()), oracle.forward(host_a[y], prm, y % self.P)), ("forward", n, y)
        self.prm, self.psis = prm, psis


def test_soak_transforms_random_sizes_batches_streams(native, oracle, gpu):
    """>= 1200 iterations (+ 200 fused products): ring degrees 2^11 .. 2^16, batches on both sides of every dispatch switch (small-batch kernels, persistent
    kernels incl. their first-polynomial path, n = 2^16 pair forward / fused inverse), three streams, forward / inverse / fused
    product; every result compared with the pinned words."""
    import torch
    rng = np.random.default_rng(2024)
    t0 = time.time()
    cases = [_Case(torch, native, oracle, gpu, 2048, P.Q60, 640, 11), _Case(torch, native, oracle, gpu, 4096, P.Q60[:3], 520, 12),
             _Case(torch, native, oracle, gpu, 8192, P.Q60, 400, 13), _Case(torch, native, oracle, gpu, 16384, P.Q60[:2], 330, 14),
             _Case(torch, native, oracle, gpu, 32768, P.Q60, 600, 15), _Case(torch, native, oracle, gpu, 65536, P.Q60[:2], 300, 16)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    pending = []                                           # (work tensor, expected tensor, label): checked in batches of launches
    iters = 0

    def flush():
        torch.cuda.synchronize()
        for w, e, label in pending:
            assert torch.equal(w, e), label
        pending.clear()

    while iters < 1200:
        c = cases[int(rng.integers(len(cases)))]
        # batch sizes: multiples of the prime count are not required -- any count <= cap
        edges = [1, 2, c.P, 100, 112, 113, 160, 161, 176, 200, 256, 257, 300, 352, 353, c.cap]
        num = int(rng.choice([e for e in edges if e <= c.cap] + [int(rng.integers(1, c.cap + 1))]))
        s = streams[int(rng.integers(3))]
        op = int(rng.integers(3))
        with torch.cuda.stream(s):
            if op == 0:                                    # forward of fresh inputs
                w = c.a[:num].clone()
                c.ctx.forward_batch(w, num, stream=s)
                pending.append((w, c.A[:num], ("forward", c.n, num)))
            elif op == 1:                                  # inverse of transformed inputs
                w = c.A[:num].clone()
                c.ctx.inverse_batch(w, num, stream=s)
                pending.append((w, c.a[:num], ("inverse", c.n, num)))
            else:                                          # forward then inverse on the same stream
                w = c.a[:num].clone()
                c.ctx.forward_batch(w, num, stream=s)
                c.ctx.inverse_batch(w, num, stream=s)
                pending.append((w, c.a[:num], ("forward+inverse", c.n, num)))
        iters += 1
        if len(pending) >= 12:
            flush()
    flush()
    # fused products (own expected words: pinned by the unfused sequence): 200 iterations
    for _ in range(200):
        c = cases[int(rng.integers(len(cases)))]
        num = int(rng.integers(1, c.cap + 1))
        s = streams[int(rng.integers(3))]
        with torch.cuda.stream(s):
            # second operand: the transformed inputs rolled by one group of P rows (row y keeps prime y % P when num is a multiple of P;
            # otherwise the rows themselves)
            bh = torch.roll(c.A[:num], shifts=c.P, dims=0).contiguous() if (num > c.P and num % c.P == 0) else c.A[:num].clone()
            w = c.a[:num].clone()
            c.ctx.polymul_batch(w, bh, num, stream=s)
            ref = c.a[:num].clone()
            c.ctx.forward_batch(ref, num, stream=s)
            c.ctx.pointwise_mul(ref, ref, bh, num, stream=s)
            c.ctx.inverse_batch(ref, num, stream=s)
            pending.append((w, ref, ("fused product", c.n, num)))
        if len(pending) >= 8:
            flush()
    flush()
    for c in cases:
        c.ctx.close()
    assert time.time() - t0 < 200


def test_soak_pair_launches_under_foreign_load(native, oracle, gpu):
    """n = 2^16 pair forward (two cooperating workgroups per polynomial, flags) while a foreign kernel on another stream holds HALF of
    the CUs for milliseconds at a time, 90 iterations, two library streams: the pair grid is not resident as a whole, partners wait for
    each other -- slow, but every word right, no watchdog abort; then the same for the 30-bit pair inverse."""
    import torch
    n = 65536
    qs = P.Q60[:2]
    c = _Case(torch, native, oracle, gpu, n, qs, 256, 21)
    cus = torch.cuda.get_device_properties(gpu).multi_processor_count
    load, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    rng = np.random.default_rng(5)
    t_plain = t_loaded = 0.0
    for it in range(90):
        num = int(rng.choice([72, 128, 200, 256]))
        loaded = it % 3 != 0
        w1, w2 = c.a[:num].clone(), c.a[:num].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if loaded:
            for _ in range(3):
                c.ctx.occupy(cus // 2, 1500, stream=load)          # half of the CUs, 1.5 ms at a time, 4.5 ms in all
        c.ctx.forward_batch(w1, num, stream=s1)
        c.ctx.forward_batch(w2, num, stream=s2)                    # (second stream: the single-workgroup form while s1 owns the flags)
        c.ctx.inverse_batch(w1, num, stream=s1)
        c.ctx.forward_batch(w1, num, stream=s1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if loaded:
            t_loaded += dt
        else:
            t_plain += dt
        assert torch.equal(w1, c.A[:num]) and torch.equal(w2, c.A[:num]), (it, num, loaded)
    c.ctx.close()
    # 30-bit path, n = 65536: pair forward (from 32 polynomials) and pair inverse (from 384), same discipline
    from test_ntt30 import PARAMS30
    q, psi, _, _, bits = PARAMS30[65536]
    prm = oracle.Params30(n, q, psi)
    num = 400
    a = np.random.default_rng(9).integers(0, q, size=(num, n), dtype=np.uint32)
    dev32 = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).to(gpu)
    d_a, d_psi, d_psiinv = dev32(a), dev32(prm.psi_tab), dev32(prm.psiinv_tab)
    ref_in = d_a.clone()
    native.forward30(d_a, n, q, prm.mu, bits, d_psi, num)
    torch.cuda.synchronize()
    A = d_a.clone()
    for y in (0, 399):
        assert np.array_equal(A[y].cpu().numpy().view(np.uint32), oracle.forward30(a[y], prm)), ("30-bit forward", y)
    c60 = native.NTTContext(2048, [P.REF_PARAMS[2048][0]], [P.REF_PARAMS[2048][1]])       # (any context: the foreign load's handle)
    for it in range(30):
        loaded = it % 2 == 1
        w = A.clone()
        if loaded:
            for _ in range(3):
                c60.occupy(cus // 2, 1500, stream=load)
        with torch.cuda.stream(s1):
            native.inverse30(w, n, q, prm.mu, bits, d_psiinv, num, stream=s1)
            native.forward30(w, n, q, prm.mu, bits, d_psi, num, stream=s1)
            native.inverse30(w, n, q, prm.mu, bits, d_psiinv, num, stream=s1)
        torch.cuda.synchronize()
        assert torch.equal(w, ref_in), ("30-bit pair round trip", it, loaded)
    c60.close()
    assert t_loaded > 0 and t_plain > 0


_PERSISTENT_CHILD = r"""
import os, sys, hashlib
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import ntt_cuda_amd as ntt, params as P
dev = torch.device("cuda", 0)
rng = np.random.default_rng(77)
out = []
for n in (2048, 4096, 8192, 16384, 32768):
    qs = P.Q60
    psis = [next(x for x in (pow(g, (q - 1) // (2 * n), q) for g in range(2, 2000)) if pow(x, n, q) == q - 1) for q in qs]
    ctx = ntt.NTTContext(n, qs, psis)
    cap = 40
    a = torch.empty((cap, n), dtype=torch.int64, device=dev); ctx.synth_splitmix(a, cap, 31)
    A = a.clone(); ctx.forward_batch(A, cap); torch.cuda.synchronize()
    bad = 0
    for it in range(30):
        num = int(rng.integers(1, cap + 1))
        w = A[:num].clone(); ctx.inverse_batch(w, num)
        f = a[:num].clone(); ctx.forward_batch(f, num)
        bad += int(not torch.equal(w, a[:num])) + int(not torch.equal(f, A[:num]))
    out.append("%d:%d:%s" % (n, bad, hashlib.sha256(A.cpu().numpy().tobytes()).hexdigest()[:16]))
    ctx.close()
print("SOAK " + " ".join(out))
"""


def test_soak_persistent_kernels_on_small_batches(native, gpu):
    """k_forward<LOGN> / k_inverse<LOGN> (and the n = 2^15 persistent kernels) forced onto batches of 1 .. 40 polynomials
    (MI355NTT_LATENCY_PATH_MAX=0: the dispatch normally gives these to the small-batch kernels): 150 forward + 150 inverse launches
    whose every workgroup runs its first-polynomial path only, all reproducing the same words -- and the same forward digest as the
    default dispatch computes for the same inputs."""
    outs = []
    for force in ("0", None):
        env = dict(os.environ)
        if force is not None:
            env["MI355NTT_LATENCY_PATH_MAX"] = force
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + _PERSISTENT_CHILD], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("SOAK ")][-1]
        fields = line.split()[1:]
        assert all(f.split(":")[1] == "0" for f in fields), line
        outs.append([f.split(":")[2] for f in fields])
    assert outs[0] == outs[1], outs


def test_soak_checked_raw_calls_follow_the_table_contents(native, oracle, gpu):
    """Reference-signature calls (forwardNTT_batch / inverseNTT_batch with the caller's tables) in checked mode, 300 iterations on two
    streams: the SAME device buffer alternately holds the tables of two different roots (rewritten in place between calls, as a caller
    that reuses an allocation would).  The cached context matches only one of them; every call must follow what the buffer holds at
    that moment -- throughput kernels for the one, literal leg for the other -- and give the oracle's words for that root."""
    import torch
    n, num = 32768, 24
    qs = P.Q60
    psis_a = P.PSI60
    psis_b = [pow(p, 3, q) for p, q in zip(P.PSI60, qs)]              # another primitive 2n-th root per prime
    prm_a, prm_b = oracle.Params(n, qs, psis_a), oracle.Params(n, qs, psis_b)
    host = oracle.synth_batch(n, num, qs, 5).reshape(num, n)
    want = {}
    for name, prm in (("a", prm_a), ("b", prm_b)):
        want[name] = _dev(torch, gpu, oracle.forward_batch(host.copy(), prm).reshape(num, n))
    tab = {k: (_dev(torch, gpu, prm.psi_tabs), _dev(torch, gpu, prm.psiinv_tabs)) for k, prm in (("a", prm_a), ("b", prm_b))}
    buf_f, buf_i = tab["a"][0].clone(), tab["a"][1].clone()
    mod = native.Moduli(qs)
    a = _dev(torch, gpu, host)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    rng = np.random.default_rng(3)
    for it in range(300):
        k = "a" if rng.integers(2) == 0 else "b"
        s = s1 if it % 2 == 0 else s2
        torch.cuda.synchronize()                            # (the table buffer is shared by both streams: one writer at a time)
        with torch.cuda.stream(s):
            buf_f.copy_(tab[k][0]); buf_i.copy_(tab[k][1])
            cnt = int(rng.integers(1, num + 1))
            w = a[:cnt].clone()
            native.forwardNTT_batch(w, n, buf_f, cnt, 4, mod, stream=s)
            f = w.clone()
            native.inverseNTT_batch(w, n, buf_i, cnt, 4, mod, stream=s)
        torch.cuda.synchronize()
        assert torch.equal(f, want[k][:cnt]), ("forward", it, k, cnt)
        assert torch.equal(w, a[:cnt]), ("round trip", it, k, cnt)
    native.raw_cache_clear()


_WATCHDOG_CHILD = r"""
import os, sys, time
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import ntt_cuda_amd as ntt, params as P
dev = torch.device("cuda", 0)
n, num = 65536, 96
qs = P.Q60[:2]
psis = [next(x for x in (pow(g, (q - 1) // (2 * n), q) for g in range(2, 2000)) if pow(x, n, q) == q - 1) for q in qs]
ctx = ntt.NTTContext(n, qs, psis)
a = torch.empty((num, n), dtype=torch.int64, device=dev); ctx.synth_splitmix(a, num, 5)
good = a.clone(); ctx.forward_batch(good, num); torch.cuda.synchronize()          # (a normal pair launch first: the reference words)
cus = torch.cuda.get_device_properties(dev).multi_processor_count
load, s1 = torch.cuda.Stream(), torch.cuda.Stream()
# (nearly) every CU held for 300 ms: a workgroup of the pair launch that becomes resident while its partner cannot gives up after 20 ms.
# How many CUs must stay free for ONE workgroup to start is the dispatcher's business (workgroups are dealt to the XCDs round-robin;
# on the development boxes one free CU starts nothing, two free CUs -- in two XCDs -- start two partner-less workgroups): try a few.
wrong, gave_up_after, held = False, 0.0, 0
faults0 = ntt.lib().mi355ntt_pair_fault_count(0)
for free in (2, 4, 1, 3, 6):
    w = a.clone(); torch.cuda.synchronize()
    ctx.occupy(cus - free, 300000, stream=load)
    time.sleep(0.02)
    t0 = time.perf_counter()
    ctx.forward_batch(w, num, stream=s1)
    torch.cuda.synchronize()
    gave_up_after = time.perf_counter() - t0
    wrong = not torch.equal(w, good)
    if wrong:
        held = cus - free
        break
# the next call that wants the pair slot reports it -- MI355NTT_EHIP, hipErrorLaunchFailure -- launches nothing and clears the condition
w2 = a.clone(); torch.cuda.synchronize()
code = hip = None
try:
    ctx.forward_batch(w2, num, stream=s1)
except ntt.NTTError as e:
    code, hip = e.code, ntt.lib().mi355ntt_last_hip_error()
torch.cuda.synchronize()
untouched = torch.equal(w2, a)
# ... and the one after it runs normally: the process still has its device context
w3 = a.clone(); torch.cuda.synchronize()
ctx.forward_batch(w3, num, stream=s1)
torch.cuda.synchronize()
print("WATCHDOG wrong=%d held=%d seconds=%.2f code=%s hip=%s untouched=%d recovered=%d faults=%d" % (wrong, held, gave_up_after, code, hip, untouched, torch.equal(w3, good),
      ntt.lib().mi355ntt_pair_fault_count(0) - faults0))
ctx.close()
"""


def test_pair_watchdog_gives_up_without_killing_the_context(native, gpu):
    """The cooperating-workgroup launch with all but a few CUs held by a foreign kernel for 300 ms and the watchdog shortened to 20 ms
    (MI355NTT_PAIR_WATCHDOG_MS): the workgroup that becomes resident never sees its partner, gives up (no trap, no hang: the launch ends
    within the foreign kernel's time, not after 256 x 20 ms -- the other workgroups find the launch marked dead and return), the next
    call reports MI355NTT_EHIP / hipErrorLaunchFailure without touching its data, and the call after that gives the right words again in
    the same process (the round-4 kernels ended in __builtin_trap, which takes the device context along)."""
    env = dict(os.environ, MI355NTT_PAIR_WATCHDOG_MS="20")
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + _WATCHDOG_CHILD], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("WATCHDOG ")][-1]
    f = dict(x.split("=") for x in line.split()[1:])
    assert f["wrong"] == "1", line                       # (the launch that gave up stored nothing / not everything: flagged, not silently right)
    assert float(f["seconds"]) < 2.0, line
    assert int(f["code"]) == native.EHIP and int(f["hip"]) == 719, line          # hipErrorLaunchFailure
    assert f["untouched"] == "1" and f["recovered"] == "1", line
    assert int(f["faults"]) >= 1, line                   # (sticky: mi355ntt_pair_fault_count still tells after the condition was cleared)
