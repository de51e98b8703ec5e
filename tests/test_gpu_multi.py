"""The cross-device paths on REAL devices (round 5).  The build box and the gpurun boxes expose one GPU, so these
tests skip there; on a node with >= 2 visible GPUs (the driver's 8-GPU run) they are the first execution of peer access, peer copies
and RCCL across devices -- each against the single-device result, word for word."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import params as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
def test_shards_object_across_real_devices(native, oracle, gpu):
    """mi355ntt_shards_* with one context per device: device-resident shards (mi355ntt_shards_transform) and the root-resident batch
    dealt out by hipMemcpyPeerAsync (mi355ntt_shards_scatter_transform_gather) against the whole-batch call on device 0."""
    import torch
    nd = _devices()
    if nd < 2:
        pytest.skip("one GPU visible: peer access / peer copies need two (covered with logical shards in test_gpu_round4.py)")
    world = min(nd, 8)
    n, qs, psis = 32768, P.Q60, P.PSI60
    num = 4 * 37 * world + 3                              # ragged: not a multiple of the prime count
    ctxs = [native.NTTContext(n, qs, psis, device=d) for d in range(world)]
    a = oracle.synth_batch(n, num, qs, 2026).reshape(num, n)
    want = native.to_device(a, "cuda:0")
    ctxs[0].forward_batch(want, num)
    torch.cuda.synchronize()
    sh = native.ShardSet(ctxs, max_polys_per_piece=64)
    full = native.to_device(a, "cuda:0")
    sh.scatter_transform_gather(0, full, num, chunks=3)          # MI355NTT_OP_FORWARD = 0
    torch.cuda.synchronize()
    assert torch.equal(full, want)
    parts = []
    for r in range(world):
        first, count = native.shard_range(num, len(qs), r, world)
        parts.append(native.to_device(a[first: first + count], "cuda:%d" % r) if count else torch.empty(0, dtype=torch.int64, device="cuda:%d" % r))
    sh.transform(0, parts, num)
    for d in range(world):
        torch.cuda.synchronize(d)
    got = torch.cat([p_.to("cuda:0").reshape(-1, n) for p_ in parts if p_.numel()])
    assert torch.equal(got, want.reshape(num, n))
    sh.transform(1, parts, num)                           # MI355NTT_OP_INVERSE: back to the input on every device
    for d in range(world):
        torch.cuda.synchronize(d)
    back = torch.cat([p_.to("cuda:0").reshape(-1, n) for p_ in parts if p_.numel()])
    assert np.array_equal(native.to_host(back).reshape(num, n), a)
    sh.close()
    for c in ctxs:
        c.close()


def _polymul_over_shards(native, oracle, devices):
    """MI355NTT_OP_POLYMUL through mi355ntt_shards_transform: every shard multiplies its polynomials with its own second operands (the
    fused product, in the NTT domain); against the whole-batch call on the first device, word for word"""
    import torch
    world = len(devices)
    n, qs, psis = 32768, P.Q60, P.PSI60
    num = 4 * 11 * world + 2                              # ragged: the last shard is shorter
    ctxs = [native.NTTContext(n, qs, psis, device=d) for d in devices]
    a = oracle.synth_batch(n, num, qs, 77).reshape(num, n)
    b = oracle.synth_batch(n, num, qs, 78).reshape(num, n)
    d0 = "cuda:%d" % devices[0]
    bh = native.to_device(b, d0)
    ctxs[0].forward_batch(bh, num)
    want = native.to_device(a, d0)
    ctxs[0].polymul_batch(want, bh, num)
    torch.cuda.synchronize()
    bh_host = native.to_host(bh).reshape(num, n)
    sh = native.ShardSet(ctxs, max_polys_per_piece=64)
    parts, bparts = [], []
    for r in range(world):
        first, count = native.shard_range(num, len(qs), r, world)
        dev = "cuda:%d" % devices[r]
        parts.append(native.to_device(a[first: first + count], dev) if count else torch.empty(0, dtype=torch.int64, device=dev))
        bparts.append(native.to_device(bh_host[first: first + count], dev) if count else torch.empty(0, dtype=torch.int64, device=dev))
    sh.transform(native.OP_POLYMUL, parts, num, bhat=bparts)
    for d in set(devices):
        torch.cuda.synchronize(d)
    got = torch.cat([p_.to(d0).reshape(-1, n) for p_ in parts if p_.numel()])
    assert torch.equal(got, want.reshape(num, n))
    sh.close()
    for c in ctxs:
        c.close()


@pytest.mark.gpu
def test_shards_polymul_across_real_devices(native, oracle, gpu):
    """round 6: the fused product across devices"""
    nd = _devices()
    if nd < 2:
        pytest.skip("one GPU visible (the same call over logical shards on one device: test_shards_polymul_logical_shards_one_device)")
    _polymul_over_shards(native, oracle, list(range(min(nd, 8))))


@pytest.mark.gpu
def test_shards_polymul_logical_shards_one_device(native, oracle, gpu):
    """... and the same code path with four logical shards on the one device this box has (what one GPU allows)"""
    _polymul_over_shards(native, oracle, [0, 0, 0, 0])


@pytest.mark.gpu
def test_bench_over_rccl_across_real_devices(native, gpu):
    """bench.py --gpus N as the driver launches it (torch.distributed.run, one rank per GPU, RCCL): the line carries the RCCL world
    size it observed and both figures of SURVEY.md 8(e) -- `value` (device-resident shards) and end_to_end (rank-0-resident batch)."""
    nd = _devices()
    if nd < 2:
        pytest.skip("one GPU visible: RCCL across devices needs two (world size 1 is covered in test_gpu_round4.py)")
    world = 2 if nd < 4 else 4
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "10", "--warmup", "3", "--batch", "256",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["collective"]["world_size_observed"] == world and line["collective"]["backend"] == "nccl"
    assert line["value"] > 0 and line["end_to_end"]["pairs_per_s"] > 0 and line["end_to_end"]["global_batch"] == world * 256
    assert 0 < line["per_rank_pairs_per_s"]["min"] <= line["per_rank_pairs_per_s"]["max"]
