"""CPU: the multi-GPU shard layer (one process per GPU, no data-path collective) with gloo, world size 2.
The per-rank transform is stood in for by the oracle (tests may use it); what is under test is the partition,
the scatter/gather plumbing and the timing reduction that bench.py uses at N > 1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import params as P


def test_shard_ranges_cover_and_respect_division():
    from ntt_cuda_amd.shard import shard_range
    for num, div, world in [(8192, 4, 8), (1024, 4, 1), (10, 4, 3), (7, 3, 2), (0, 4, 2), (4, 4, 8), (33, 16, 2)]:
        seen = 0
        for r in range(world):
            s, c = shard_range(num, div, r, world)
            assert s == seen and s % div == 0                      # contiguous, prime index preserved: y % div == (y - s) % div
            if r < world - 1:
                assert c % div == 0
            seen += c
        assert seen == num
    with pytest.raises(ValueError):
        shard_range(8, 0, 0, 1)


def test_c_abi_shard_range_is_the_same_rule(native):
    """mi355ntt_shard_range (the single-process multi-device driver's partition, csrc/shard.cpp) = ntt_cuda_amd.shard.shard_range
    (the one-process-per-GPU form) on every case, and it rejects the same bad arguments (host only: no GPU)."""
    from ntt_cuda_amd.shard import shard_range
    rng = np.random.default_rng(11)
    cases = [(8192, 4, 8), (1024, 4, 1), (10, 4, 3), (7, 3, 2), (0, 4, 2), (4, 4, 8), (33, 16, 2), (1023, 4, 8), (8192, 1, 8)]
    cases += [(int(rng.integers(0, 20000)), int(rng.integers(1, 17)), int(rng.integers(1, 17))) for _ in range(200)]
    for num, div, world in cases:
        for r in range(world):
            assert native.shard_range(num, div, r, world) == shard_range(num, div, r, world), (num, div, r, world)
    for bad in [(8, 0, 0, 1), (8, 4, 1, 1), (8, 4, 0, 0)]:
        with pytest.raises(native.NTTError):
            native.shard_range(*bad)
    # creation checks its arguments before it touches a device
    import ctypes
    h = ctypes.c_void_p()
    assert native.lib().mi355ntt_shards_create(ctypes.byref(h), None, 2, 0) == native.EINVAL
    assert native.lib().mi355ntt_shards_create(None, None, 0, 0) == native.EINVAL
    assert native.lib().mi355ntt_shards_destroy(None) == native.OK


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num, n, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (os.path.join(root, "ntt-cuda_amd"), os.path.join(root, "oracle"), here):
        sys.path.insert(0, p)
    import oracle_py as oracle
    from ntt_cuda_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    qs = [P.REF_PARAMS_4096_58BIT[0], P.REF_PARAMS[4096][0], P.EDGE_PRIMES[59][0]]
    psis = [P.REF_PARAMS_4096_58BIT[1], P.REF_PARAMS[4096][1], P.EDGE_PRIMES[59][1][4096]]
    prm = oracle.Params(n, qs, psis)
    full = None
    if rank == 0:
        full = torch.from_numpy(oracle.synth_batch(n, num, qs, 1).view(np.int64))
    local = shard.scatter_batch(full, num, n, len(qs), src=0)
    start, count = shard.shard_range(num, len(qs), rank, world)
    assert tuple(local.shape) == (count, n) and start % len(qs) == 0
    # the shard is transformed with the SAME rule "polynomial y -> prime y % division" because start % division == 0
    if count:
        res = oracle.forward_batch(local.numpy().view(np.uint64), prm, division=len(qs)).reshape(count, n)
        local = torch.from_numpy(res.view(np.int64))
    dist.barrier()
    got = shard.gather_batch(local, num, n, len(qs), dst=0)
    worst = shard.max_over_ranks(0.25 + rank)
    assert worst == 0.25 + world - 1
    if rank == 0:
        want = oracle.forward_batch(full.numpy().view(np.uint64), prm, division=len(qs)).reshape(num, n)
        np.save(out, np.array([int(np.array_equal(got.numpy().view(np.uint64), want))]))
    dist.destroy_process_group()


@pytest.mark.parametrize("num", [9, 7])
def test_scatter_transform_gather_world2(tmp_path, num):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, _free_port(), num, 4096, out), nprocs=2, join=True)
    assert np.load(out)[0] == 1


def _worker_e2e(rank, world, port, num, n, chunks, out, inplace=False):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (os.path.join(root, "ntt-cuda_amd"), os.path.join(root, "oracle"), here):
        sys.path.insert(0, p)
    import oracle_py as oracle
    from ntt_cuda_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    qs = [P.REF_PARAMS_4096_58BIT[0], P.REF_PARAMS[4096][0], P.EDGE_PRIMES[59][0]]
    psis = [P.REF_PARAMS_4096_58BIT[1], P.REF_PARAMS[4096][1], P.EDGE_PRIMES[59][1][4096]]
    prm = oracle.Params(n, qs, psis)
    full = torch.from_numpy(oracle.synth_batch(n, num, qs, 3).view(np.int64)) if rank == 0 else None
    calls = []

    def transform(piece, count):                      # stands in for ctx.forward_batch(piece, count) on a GPU
        calls.append(count)
        res = oracle.forward_batch(piece.numpy().view(np.uint64), prm, division=len(qs)).reshape(count, n)
        piece.copy_(torch.from_numpy(res.view(np.int64)))

    keep = full.clone() if rank == 0 else None
    got = shard.scatter_transform_gather(full, num, n, len(qs), transform, chunks=chunks, src=0, inplace=inplace)
    assert sum(calls) == shard.shard_range(num, len(qs), rank, world)[1]
    if rank == 0:
        assert (got.data_ptr() == full.data_ptr()) == inplace          # in place: the root holds the batch once
        want = oracle.forward_batch(keep.numpy().view(np.uint64), prm, division=len(qs)).reshape(num, n)
        np.save(out, np.array([int(np.array_equal(got.numpy().view(np.uint64), want))]))
    dist.destroy_process_group()


@pytest.mark.parametrize("num,chunks", [(24, 3), (7, 4), (2, 2)])
def test_pipelined_scatter_transform_gather_world2(tmp_path, num, chunks):
    """the chunked end-to-end pipeline (pieces in flight both ways while one is transformed) returns what one call on the
    whole batch returns, also for ragged batches and when a rank gets nothing"""
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker_e2e, args=(2, _free_port(), num, 4096, chunks, out), nprocs=2, join=True)
    assert np.load(out)[0] == 1


@pytest.mark.parametrize("num,chunks", [(24, 3), (7, 4)])
def test_pipelined_scatter_transform_gather_in_place_world2(tmp_path, num, chunks):
    """inplace=True: the results return into the root's own batch tensor (no second [num, n] buffer at the root)"""
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker_e2e, args=(2, _free_port(), num, 4096, chunks, out, True), nprocs=2, join=True)
    assert np.load(out)[0] == 1


def test_piece_plan():
    from ntt_cuda_amd.shard import _pieces
    for count, div, chunks in [(24, 3, 4), (7, 3, 4), (2, 3, 2), (1024, 4, 4), (5, 4, 8)]:
        ps = _pieces(count, div, chunks)
        assert sum(c for _, c in ps) == count and all(s % div == 0 for s, _ in ps)
        assert [s for s, _ in ps] == sorted(s for s, _ in ps) and len(ps) <= chunks
