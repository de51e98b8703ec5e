"""Batch sharding across the GPUs of one node.

The reference is single-GPU; every polynomial of a batch is independent (ntt_60bit.cuh:391,404,422: blockIdx.y
only selects the data offset y*n and the modulus y % division).  So ranks own contiguous ranges of WHOLE
polynomials, range boundaries at multiples of `division` (polynomial y keeps prime y % division inside its shard),
the small context is replicated, and the transforms need no collective.  RCCL (torch.distributed backend "nccl";
"gloo" in the CPU tests) is used only to scatter a root-held batch and gather the results, as point-to-point
sends so shard sizes may differ.
"""
import torch
import torch.distributed as dist


def shard_range(num, division, rank, world):
    """(first polynomial, count) of `rank`'s shard: groups of `division` polynomials dealt as evenly as possible."""
    if division <= 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    groups, tail = divmod(num, division)
    base, extra = divmod(groups, world)
    g0 = rank * base + min(rank, extra)
    gcount = base + (1 if rank < extra else 0)
    start, count = g0 * division, gcount * division
    if rank == world - 1:
        count += tail                       # a ragged tail (num % division polynomials) stays with the last rank
    return start, count


def scatter_batch(full, num, n, division, src=0, group=None, device=None, dtype=torch.int64):
    """Root `src` holds `full` ([num, n]); every rank returns its own [count, n] shard."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    start, count = shard_range(num, division, rank, world)
    if rank == src:
        full = full.reshape(num, n)
        device = full.device
        ops = []
        for r in range(world):
            if r == src:
                continue
            s, c = shard_range(num, division, r, world)
            if c:
                ops.append(dist.P2POp(dist.isend, full[s:s + c].contiguous(), r, group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        local = full[start:start + count].clone()
        for q in reqs:
            q.wait()
        return local
    local = torch.empty((count, n), dtype=dtype, device=device)
    if count:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.irecv, local, src, group)]):
            q.wait()
    return local


def gather_batch(local, num, n, division, dst=0, group=None):
    """Inverse of scatter_batch: root `dst` returns the [num, n] batch, other ranks None."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    start, count = shard_range(num, division, rank, world)
    if rank == dst:
        full = torch.empty((num, n), dtype=local.dtype, device=local.device)
        full[start:start + count] = local.reshape(count, n)
        ops, bufs = [], []
        for r in range(world):
            if r == dst:
                continue
            s, c = shard_range(num, division, r, world)
            if c:
                ops.append(dist.P2POp(dist.irecv, full[s:s + c], r, group))
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        return full
    if count:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.reshape(count, n).contiguous(), dst, group)]):
            q.wait()
    return None


def max_over_ranks(value, device=None, group=None):
    """Max of a python float over ranks (the bench's timing reduction)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
