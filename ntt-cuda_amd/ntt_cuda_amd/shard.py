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


def _pieces(count, division, chunks):
    """split `count` polynomials into up to `chunks` contiguous pieces whose starts are multiples of `division`"""
    groups, tail = divmod(count, division)
    chunks = max(1, min(chunks, groups if groups else 1))
    base, extra = divmod(groups, chunks)
    out, g = [], 0
    for c in range(chunks):
        gc = base + (1 if c < extra else 0)
        out.append([g * division, gc * division])
        g += gc
    out[-1][1] += tail
    return [(s, c) for s, c in out if c]


def scatter_transform_gather(full, num, n, division, transform, chunks=4, src=0, group=None, device=None, dtype=torch.int64,
                             inplace=False):
    """End to end from a root-held batch (report 2): root `src` deals the shards out in `chunks` pieces
    per rank, every rank transforms piece k in place with `transform(piece_tensor, count)` while piece k + 1 is arriving
    and piece k - 1 is on its way back, and `src` returns the assembled [num, n] result (other ranks None).

    Pieces start at multiples of `division`, so `transform` sees polynomial y of a piece with prime y % division -- the
    same call as on the whole batch.  At 8 GPUs the transfers dominate (256 MiB per peer over one xGMI link each way vs
    < 1 ms of transforms), so this is what a caller with a host- or root-resident batch should expect; the compute-only
    figure is bench.py's default.

    inplace=True: the results come back into `full` itself (returned) instead of a second [num, n] tensor -- at BASELINE
    configs[3] (8192 polynomials, 2 GiB) the root then holds the batch once, not twice.  Safe by construction: the rows of
    piece k are sent in round k (waited for at the end of that round) and received back in round k + 2."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    plan = {r: shard_range(num, division, r, world) for r in range(world)}
    pieces = {r: [(plan[r][0] + s, c) for s, c in _pieces(plan[r][1], division, chunks)] if plan[r][1] else [] for r in range(world)}
    depth = max(len(p) for p in pieces.values()) if pieces else 0
    if rank == src:
        full = full.reshape(num, n)
        out = full if inplace else torch.empty_like(full)
        for k in range(depth + 2):
            ops = []
            for r in range(world):                                   # piece k goes out, piece k - 2 comes back
                if r == src:
                    continue
                if k < len(pieces[r]):
                    s, c = pieces[r][k]
                    ops.append(dist.P2POp(dist.isend, full[s:s + c].contiguous(), r, group))
                if 0 <= k - 2 < len(pieces[r]):
                    s, c = pieces[r][k - 2]
                    ops.append(dist.P2POp(dist.irecv, out[s:s + c], r, group))
            reqs = dist.batch_isend_irecv(ops) if ops else []
            if 0 <= k - 1 < len(pieces[src]):                        # the root's own piece k - 1 meanwhile
                s, c = pieces[src][k - 1]
                piece = out[s:s + c]                                   # contiguous rows of the result: transform in place there
                if not inplace:
                    piece.copy_(full[s:s + c])
                transform(piece, c)
            for q in reqs:
                q.wait()
        return out
    mine = pieces[rank]
    bufs = [torch.empty((c, n), dtype=dtype, device=device) for _, c in mine]
    for k in range(depth + 2):
        ops = []
        if k < len(mine):
            ops.append(dist.P2POp(dist.irecv, bufs[k], src, group))
        if 0 <= k - 2 < len(mine):
            ops.append(dist.P2POp(dist.isend, bufs[k - 2], src, group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        # piece k - 1 arrived in round k - 1 (waited for at the end of that round): transform it while k arrives and k - 2
        # leaves.  No host synchronisation: the transform is enqueued on the current stream, and the backend orders the send
        # of round k + 1 behind it (RCCL work is launched on a stream that waits for the current stream's tail; work.wait()
        # makes the current stream wait for the transfer) -- the hand-off is stream-ordered, event based.
        if 0 <= k - 1 < len(mine):
            transform(bufs[k - 1], mine[k - 1][1])
        for q in reqs:
            q.wait()
    return None
