"""GPU, round 4: the single-process multi-device driver of the C ABI (mi355ntt_shards_*) with logical shards on the one device, RCCL
initialised and used once under test (torch.distributed backend "nccl" at world size 1: barrier, MAX all-reduce, a self send/recv,
scatter_transform_gather), the stand-alone element-wise wrappers of poly_arithmetic.cuh:312-352, per-prime literal routing, and the
61-bit kernel class."""
import os
import subprocess
import sys

import numpy as np
import pytest

import params as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _residues(torch, gpu, num, n, qs, seed):
    g = torch.Generator(device=gpu).manual_seed(seed)
    qcol = torch.tensor(np.array(qs, dtype=np.uint64).view(np.int64), device=gpu)[torch.arange(num, device=gpu) % len(qs)].unsqueeze(1)
    a = torch.randint(0, 1 << 60, (num, n), dtype=torch.int64, device=gpu, generator=g)
    return torch.where(a >= qcol, a - qcol, a).contiguous()


@pytest.mark.parametrize("num,world", [(1024, 4), (1023, 8), (6, 4)])
def test_shardset_logical_shards_equal_whole_batch(native, oracle, gpu, num, world):
    """SURVEY 8(e) through the C ABI: `world` lanes (one context each, all on cuda:0 -- logical shards) give the words of the
    whole-batch call, for device-resident shards (forward, inverse, fused product) and for a root-resident batch dealt out in pieces
    (more pieces than staging buffers: the ring wraps), also with a ragged tail and with empty shards."""
    import torch
    n, qs, psis = 32768, P.Q60, P.PSI60
    ctxs = [native.NTTContext(n, qs, psis) for _ in range(world)]
    sh = native.ShardSet(ctxs, max_polys_per_piece=64)
    assert sh.world == world
    a, b = _residues(torch, gpu, num, n, qs, 7 + num), _residues(torch, gpu, num, n, qs, 8 + num)
    whole_f = a.clone()
    ctxs[0].forward_batch(whole_f, num)
    bh = b.clone()
    ctxs[0].forward_batch(bh, num)
    whole_m = a.clone()
    ctxs[0].polymul_batch(whole_m, bh, num)
    # device-resident shards: separate tensors per lane
    ranges = [native.shard_range(num, 4, r, world) for r in range(world)]
    assert sum(c for _, c in ranges) == num
    parts = [a[s:s + c].clone() if c else torch.empty((0, n), dtype=torch.int64, device=gpu) for s, c in ranges]
    sh.transform(native.OP_FORWARD, parts, num)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts), whole_f)
    sh.transform(native.OP_INVERSE, parts, num)
    assert torch.equal(torch.cat(parts), a)               # (the join is stream-ordered: torch.cat runs on the current stream)
    bparts = [bh[s:s + c].clone() if c else torch.empty((0, n), dtype=torch.int64, device=gpu) for s, c in ranges]
    sh.transform(native.OP_POLYMUL, parts, num, bhat=bparts)
    assert torch.equal(torch.cat(parts), whole_m)
    # root-resident batch: scatter / transform / gather in place, pieces of at most 64 polynomials
    full = a.clone()
    sh.scatter_transform_gather(native.OP_FORWARD, full, num, chunks=5)
    assert torch.equal(full, whole_f)
    sh.scatter_transform_gather(native.OP_INVERSE, full, num, chunks=2)
    assert torch.equal(full, a)
    for _ in range(3):                                    # back to back: the staging ring is reused across calls
        sh.scatter_transform_gather(native.OP_FORWARD_INVERSE, full, num, chunks=3)
    assert torch.equal(full, a)
    # a sample against the oracle
    prm = oracle.Params(n, qs, psis)
    for y in (0, num - 1):
        assert np.array_equal(native.to_host(whole_f[y].contiguous()), oracle.forward(native.to_host(a[y].contiguous()), prm, y % 4))
    with pytest.raises(native.NTTError):
        sh.scatter_transform_gather(native.OP_POLYMUL, full, num)
    sh.close()
    for c in ctxs:
        c.close()


_NCCL_CHILD = r"""
import os, sys, socket
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import ntt_cuda_amd as ntt
from ntt_cuda_amd import shard
from bench import Q60, PSI60
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
dist.barrier()
assert shard.max_over_ranks(1.25, device=dev) == 1.25                      # the bench's MAX reduction, on the GPU through RCCL
x = torch.arange(1 << 16, dtype=torch.int64, device=dev); y = torch.zeros_like(x)
for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, x, 0), dist.P2POp(dist.irecv, y, 0)]):     # RCCL send/recv (to itself)
    q.wait()
torch.cuda.synchronize()
assert torch.equal(x, y)
n, num = 32768, 64
ctx = ntt.NTTContext(n, Q60, PSI60)
full = torch.empty((num, n), dtype=torch.int64, device=dev); ctx.synth_splitmix(full, num, 1)
ref = full.clone()
def tf(piece, count):
    ctx.forward_batch(piece, count); ctx.inverse_batch(piece, count)
got = shard.scatter_transform_gather(full, num, n, 4, tf, chunks=4, src=0, device=dev, inplace=True)
torch.cuda.synchronize()
assert got.data_ptr() == full.data_ptr() and torch.equal(full, ref)
loc = shard.scatter_batch(full, num, n, 4, src=0, device=dev)
back = shard.gather_batch(loc, num, n, 4, dst=0)
assert torch.equal(back, ref)
dist.barrier()
dist.destroy_process_group()
print("NCCL-OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
"""


def test_rccl_is_initialised_and_used_once_world_size_1(native, gpu):
    """No multi-GPU node is available to this build, so RCCL cannot move a shard between GPUs here; what one GPU allows is run:
    torch.distributed backend "nccl" (= RCCL) at world size 1 -- process group on cuda:0, barrier, the bench's MAX all-reduce, a
    self send/recv through batch_isend_irecv, shard.scatter_transform_gather / scatter_batch / gather_batch on CUDA tensors."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + _NCCL_CHILD], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "NCCL-OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_per_prime_literal_routing_on_the_kat1_moduli(native, oracle, gpu):
    """decryption_test.cu's own primes: 0 and 2 are Barrett-exact, prime 1 (68719230977) is not.  The context routes per prime
    (mi355ntt_ctx_uses_literal_kernels == 2): polynomials of prime 1 through the reference's own butterflies, the others through the lazy
    ones (since round 6 in one single-pass launch, kernel class 0) -- and every word is the oracle's (the reference's, including its
    non-canonical outputs on prime 1), for large batches, ragged tails, sub-ranges of the primes, two streams at once, and the fused-product entry points."""
    import torch
    from test_barrett_exactness import KAT_Q, KAT_PSI
    n = 4096
    ctx = native.NTTContext(n, KAT_Q, KAT_PSI)
    assert ctx.literal_routing == 2 and ctx.uses_literal_kernels
    assert [native.barrett_is_exact(q) for q in KAT_Q] == [True, False, True]
    prm = oracle.Params(n, KAT_Q, KAT_PSI)
    for num in (3, 1000, 1537):                       # 256 groups of 3 per chunk: 1 chunk, 2 chunks (ragged), 3 chunks (ragged)
        # ternary-like inputs hit the inexact case often (operand q - 1): oracle.bfv_sample's ternary rows, tiled, plus uniform rows
        tern = oracle.bfv_sample(KAT_Q, n, 7)["ternary"]
        a = oracle.synth_batch(n, num, KAT_Q, 100 + num).reshape(num, n)
        for y in range(0, num, 5):
            a[y] = tern[y % 3]
        want_f = oracle.forward_batch(a.copy(), prm).reshape(num, n)
        if num >= 1000:
            assert (want_f[1::3] >= KAT_Q[1]).any()      # the reference's non-canonical words are part of the expectation
        d = native.to_device(a)
        ctx.forward_batch(d, num)
        torch.cuda.synchronize()
        assert np.array_equal(native.to_host(d).reshape(num, n), want_f), ("forward", num)
        ctx.inverse_batch(d, num)
        assert np.array_equal(native.to_host(d).reshape(num, n), oracle.inverse_batch(want_f.copy(), prm).reshape(num, n)), ("inverse", num)
    # two streams at once on the same context (nothing is shared between them)
    num = 900
    a = oracle.synth_batch(n, num, KAT_Q, 55).reshape(num, n)
    b = oracle.synth_batch(n, num, KAT_Q, 56).reshape(num, n)
    def chain(x):          # the same seven transforms in the reference's arithmetic (on prime 1 inverse(forward(x)) need not be x)
        w = x.copy()
        for _ in range(3):
            w = oracle.inverse_batch(oracle.forward_batch(w, prm), prm)
        return oracle.forward_batch(w, prm).reshape(num, n)

    want_a, want_b = chain(a), chain(b)
    da, db = native.to_device(a), native.to_device(b)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.forward_batch(da, num, stream=s1); ctx.forward_batch(db, num, stream=s2)
        ctx.inverse_batch(da, num, stream=s1); ctx.inverse_batch(db, num, stream=s2)
    ctx.forward_batch(da, num, stream=s1); ctx.forward_batch(db, num, stream=s2)
    torch.cuda.synchronize()
    assert np.array_equal(native.to_host(da).reshape(num, n), want_a) and np.array_equal(native.to_host(db).reshape(num, n), want_b)
    # captured into a hipGraph and replayed
    num = 300
    a = oracle.synth_batch(n, num, KAT_Q, 91).reshape(num, n)
    want = oracle.inverse_batch(oracle.forward_batch(a.copy(), prm), prm).reshape(num, n)
    d = native.to_device(a)
    src = native.to_device(a)
    torch.cuda.synchronize()
    g, cs = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.graph(g, stream=cs):
        ctx.forward_batch(d, num)
        ctx.inverse_batch(d, num)
    for _ in range(2):
        d.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(native.to_host(d).reshape(num, n), want), "mixed context, graph replay"
    # single-polynomial entry points pick the kernel class of their prime; a division that covers only exact primes runs fast kernels
    for i in range(3):
        x = oracle.synth_batch(n, 1, [KAT_Q[i]], 9 + i)[0]
        dx = native.to_device(x)
        ctx.forward(dx, i)
        assert np.array_equal(native.to_host(dx), oracle.forward(x, prm, i)), i
    one = oracle.synth_batch(n, 40, KAT_Q[:1], 77).reshape(40, n)
    d1 = native.to_device(one)
    ctx.forward_batch(d1, 40, division=1)
    prm0 = oracle.Params(n, KAT_Q[:1], KAT_PSI[:1])
    assert np.array_equal(native.to_host(d1).reshape(40, n), oracle.forward_batch(one.copy(), prm0).reshape(40, n))
    # fused product = the reference's three-step sequence on every prime
    num = 301
    a = oracle.synth_batch(n, num, KAT_Q, 61).reshape(num, n)
    bh = oracle.synth_batch(n, num, KAT_Q, 62).reshape(num, n)
    da = native.to_device(a)
    ctx.polymul_batch(da, native.to_device(bh), num)
    want = oracle.inverse_batch(oracle.pointwise_batch(oracle.forward_batch(a.copy(), prm), bh, prm), prm)
    assert np.array_equal(native.to_host(da).reshape(-1), want.reshape(-1))
    ctx.close()


def test_elementwise_wrappers_match_the_reference_arithmetic(native, oracle, gpu):
    """poly_arithmetic.cuh:312-352 -- poly_add_device, poly_mul_int_t, poly_sub_device, poly_negate_device,
    poly_add_integer_device -- against numpy restatements of the reference kernels (:128-179,334-338), including their quirks:
    `>` in the additions (a sum equal to q stays q), poly_sub never subtracting, 0 -> 0 in the negation, the 32-bit mask of mod_t."""
    import torch
    rng = np.random.default_rng(5)
    for n, q in ((2048, P.REF_PARAMS[2048][0]), (32768, P.Q60[0]), (4096, P.EDGE_PRIMES[61][0])):
        a = rng.integers(0, q, size=n, dtype=np.uint64)
        b = rng.integers(0, q, size=n, dtype=np.uint64)
        a[:6] = [0, 1, q - 1, q - 1, 5, 7]
        b[:6] = [0, q - 1, 1, q - 1, 7, 5]                      # sums 0, q, q, 2q - 2; differences both ways
        s = torch.cuda.current_stream()
        d = native.to_device(a); native.poly_add_device(d, native.to_device(b), n, s, q)
        got = native.to_host(d)
        assert np.array_equal(got, oracle.poly_add(a, b, q)) and got[1] == q and got[2] == q        # `>`: q is left alone
        d = native.to_device(a); native.poly_sub_device(d, native.to_device(b), n, s, q)
        got = native.to_host(d)
        assert np.array_equal(got, oracle.poly_sub(a, b, q)) and got[4] == 5 + q and got[5] == 7   # (never a - b)
        d = native.to_device(a); native.poly_negate_device(d, n, s, q)
        got = native.to_host(d)
        assert np.array_equal(got, oracle.poly_negate(a, q)) and got[0] == 0 and got[1] == q - 1
        for k in (0, 1, q - 1, 12345678901234567):
            d = native.to_device(a); native.poly_add_integer_device(d, k, n, s, q)
            assert np.array_equal(native.to_host(d), oracle.poly_add_integer(a, k, q)), k
        for t, k in ((1024, 3), (1 << 16, q - 5), (1 << 40, 0x123456789abcdef)):                   # t - 1 wider than 32 bits: truncated
            d = native.to_device(a); native.poly_mul_int_t(d, k, n, s, t)
            assert np.array_equal(native.to_host(d), oracle.poly_mul_int_t(a, k, t)), (t, k)
    # odd word counts and tails (the wrappers take any n; the reference launches n / 256 blocks)
    a = rng.integers(0, 1 << 50, size=1030, dtype=np.uint64)
    d = native.to_device(a)[:1029]
    native.poly_negate_device(d, 1029, torch.cuda.current_stream(), 1 << 50)
    assert np.array_equal(native.to_host(d), oracle.poly_negate(a[:1029], 1 << 50))
    L = native.lib()
    assert L.mi355ntt_poly_add_raw(None, None, 16, None, 17) == native.EINVAL
    assert L.mi355ntt_poly_negate_raw(native.vp(4), 16, None, 17) == native.EINVAL                 # not even a word boundary
    # pointers that are word- but not 16-byte aligned (a + odd offset) are transformed, as by the reference's kernels:
    # the one-word-per-lane form of every wrapper against the same restatements
    q = P.Q60[0]
    a = rng.integers(0, q, size=2050, dtype=np.uint64)
    b = rng.integers(0, q, size=2050, dtype=np.uint64)
    s = torch.cuda.current_stream()
    for off_a, off_b in ((1, 0), (0, 1), (1, 1)):
        da, db = native.to_device(a), native.to_device(b)
        native.poly_add_device(da[off_a:], db[off_b:], 2048, s, q)
        got = native.to_host(da)
        assert np.array_equal(got[off_a: off_a + 2048], oracle.poly_add(a[off_a: off_a + 2048], b[off_b: off_b + 2048], q)), (off_a, off_b)
        assert np.array_equal(got[:off_a], a[:off_a]) and np.array_equal(got[off_a + 2048:], a[off_a + 2048:])      # nothing outside
        da = native.to_device(a)
        native.poly_sub_device(da[off_a:], db[off_b:], 2048, s, q)
        assert np.array_equal(native.to_host(da)[off_a: off_a + 2048], oracle.poly_sub(a[off_a: off_a + 2048], b[off_b: off_b + 2048], q))
    da = native.to_device(a)
    native.poly_negate_device(da[1:], 2048, s, q)
    assert np.array_equal(native.to_host(da)[1:2049], oracle.poly_negate(a[1:2049], q))
    da = native.to_device(a)
    native.poly_add_integer_device(da[1:], 77, 2048, s, q)
    assert np.array_equal(native.to_host(da)[1:2049], oracle.poly_add_integer(a[1:2049], 77, q))
    da = native.to_device(a)
    native.poly_mul_int_t(da[1:], 3, 2048, s, 1024)
    assert np.array_equal(native.to_host(da)[1:2049], oracle.poly_mul_int_t(a[1:2049], 3, 1024))


def test_configs3_global_batch_8192_as_8_shards_on_one_gpu(native, oracle, gpu):
    """BASELINE configs[3] at its FULL global batch (n = 32768, 4-prime RNS, 8192 polynomials = 2 GiB), resident on the one GPU: the
    whole-batch call against the 8-shard decomposition the 8-GPU run would use (mi355ntt_shards_transform over eight contexts), word
    for word over all 2^28 coefficients, the round trip, and a sample of polynomials against the oracle."""
    import torch
    n, qs, psis, num, world = 32768, P.Q60, P.PSI60, 8192, 8
    ctxs = [native.NTTContext(n, qs, psis) for _ in range(world)]
    a = torch.empty((num, n), dtype=torch.int64, device=gpu)
    ctxs[0].synth_splitmix(a, num, 1)
    whole = a.clone()
    ctxs[0].forward_batch(whole, num)
    sh = native.ShardSet(ctxs)
    shards = a.clone()
    parts = [shards[r * 1024:(r + 1) * 1024] for r in range(world)]
    assert [native.shard_range(num, 4, r, world) for r in range(world)] == [(r * 1024, 1024) for r in range(world)]
    sh.transform(native.OP_FORWARD, parts, num)
    assert torch.equal(shards, whole)
    sh.transform(native.OP_INVERSE, parts, num)
    assert torch.equal(shards, a)
    prm = oracle.Params(n, qs, psis)
    host = oracle.synth_batch(n, num, qs, 1).reshape(num, n)
    for y in (0, 1023, 1024, 4097, 8191):
        assert np.array_equal(native.to_host(a[y].contiguous()), host[y])
        assert np.array_equal(native.to_host(whole[y].contiguous()), oracle.forward(host[y], prm, y % 4)), y
    sh.close()
    for c in ctxs:
        c.close()


def test_bench_collective_bracket_over_rccl_world_size_1(native, gpu):
    """bench.py's N > 1 bracket -- init_process_group("nccl"), barrier + synchronize on both sides of the timed region, the MAX
    all-reduce of the elapsed time, barrier + destroy_process_group before the CPU leg, the --end-to-end leg -- executed over RCCL on the
    one GPU (MI355NTT_BENCH_FORCE_PG=1 joins a process group at world size 1).  Not a scaling number; the line must come out whole."""
    import json
    env = dict(os.environ, MI355NTT_BENCH_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--batch", "256", "--no-extras",
                        "--no-cpu-baseline", "--end-to-end"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["scaling"] == "weak"
    assert "error" not in out.get("end_to_end", {}), out.get("end_to_end")
    # round 6: the slowest / fastest rank from the same all-reduce as the contract's MAX, the settled flag and the rounds beside `value`
    assert out["per_rank_pairs_per_s"]["min"] <= out["per_rank_pairs_per_s"]["max"] and abs(out["per_rank_pairs_per_s"]["min"] - out["value"]) < 1e-6 * out["value"]
    assert isinstance(out["settled"], bool) and out["rounds"]["pairs_per_s_median"] > 0 and out["roofline"]["rounds_pairs_per_s"]["median"] > 0
