"""Partition of a batch of frame-sets across ranks, and the final depth-map gather.

The decode itself is pixel-independent, so ranks never
exchange data while computing; the only collective is the gather of finished
depth maps (RCCL over xGMI on GPUs -- torch.distributed backend "nccl" -- or
gloo on CPU in the tests).  Two ways to cut the work, both exact:
  * by frame-set: rank r decodes whole frame-sets [lo, hi);
  * by row-tile:  rank r decodes rows [lo, hi) of every frame-set (the split
    BASELINE.json's north_star words); the tile's row_offset keeps (v - cy) of
    R/CCalculation.cpp:160 referring to the full-frame row.
"""
import torch
import torch.distributed as dist


def split_range(n, world, rank):
    """Contiguous, balanced [lo, hi) of n units for `rank` of `world` (earlier ranks get the remainder)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def row_tile_spec(spec, world, rank):
    """The spec of this rank's row tile (height and row_offset adjusted)."""
    lo, hi = split_range(spec["height"], world, rank)
    tile = dict(spec)
    tile["height"] = hi - lo
    tile["row_offset"] = spec.get("row_offset", 0) + lo
    return tile, lo, hi


def gather_depth(local, dst=0, group=None, all_ranks=False):
    """Gathers per-rank depth tensors along dim 0.

    local: [n_local, ...] tensor (same trailing shape on every rank; n_local may
    differ by one between ranks).  Returns the concatenation on `dst` (None on
    other ranks), or on every rank when all_ranks=True.  One collective call.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    counts[rank] = local.shape[0]
    dist.all_reduce(counts, group=group)
    counts = [int(c) for c in counts.tolist()]
    if len(set(counts)) == 1:
        # equal shards: a single all_gather_into_tensor / gather of one flat buffer
        if all_ranks:
            out = torch.empty((world * counts[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(out, local.contiguous(), group=group)
            return out
        bufs = None
        if rank == dst:
            out = torch.empty((world * counts[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            bufs = list(out.split(counts[0], dim=0))
        dist.gather(local.contiguous(), bufs, dst=dst, group=group)
        return out if rank == dst else None
    # ragged shards: pad to the largest, gather, trim
    m = max(counts)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    gathered = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(gathered, pad, group=group)
    if not all_ranks and rank != dst:
        return None
    return torch.cat([g[:c] for g, c in zip(gathered, counts)], dim=0)


# ---- the shard table and the reassembling gather ---------------------------------------------------------------------
# A shard = (set0, n_sets, row0, rows): the rows [row0, row0+rows) of the frame-sets [set0, set0+n_sets) of a batch whose
# full result is [total_sets][height][width].  The same table drives the native RCCL gather (slx_gather_depth /
# slx_decode_gather in include/slx.h, api.Comm) and the torch.distributed one below, which exists for CPU (gloo) tests and
# one-GPU rehearsals: it posts the messages libslx's own planner lists (slx_gather_plan_ex), in either gather shape.

def shards_by_frameset(total_sets, world, height):
    out = []
    for r in range(world):
        lo, hi = split_range(total_sets, world, r)
        out.append((lo, hi - lo, 0, height))
    return out


def shards_by_rows(total_sets, world, height):
    out = []
    for r in range(world):
        lo, hi = split_range(height, world, r)
        out.append((0, total_sets, lo, hi - lo))
    return out


def total_sets(shards):
    return max((s0 + n for s0, n, _, _ in shards), default=0)


def gather_shards(local, shards, height, width, dst=0, group=None, shape="in_place", chunk=None):
    """local: this rank's shard, [n_sets, rows, width] (contiguous).  Returns the reassembled [total_sets, height, width]
    tensor on rank `dst` (None elsewhere), or on every rank when dst is None.  Row tiles land at their row offset of every
    frame-set; ragged tile heights and ragged set counts are fine.

    The messages are the NATIVE gather's: every group is planned by libslx's slx_gather_plan_ex (csrc/slx_comm.cpp: plan_range,
    the code the RCCL path posts from) and posted here through torch.distributed point to point, so a gloo run exercises the same
    schedule the GPUs would see.  shape: "in_place" (one message per (peer, frame-set), landing in place) or "staged" (one message
    per (peer, chunk) into a staging buffer, then the planner's row scatter); chunk: frame-sets per group (None: all in one)."""
    from . import api
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(shards) == world
    set0, n, row0, rows = shards[rank]
    assert tuple(local.shape) == (n, rows, width), (tuple(local.shape), shards[rank], width)
    local = local.contiguous()
    flat_local = local.reshape(-1)
    receives = dst is None or dst == rank
    root = -1 if dst is None else dst
    full = flat_full = None
    if receives:
        full = torch.zeros((total_sets(shards), height, width), dtype=local.dtype, device=local.device)
        full[set0:set0 + n, row0:row0 + rows] = local                     # this rank's own shard (gather_range's self copy)
        flat_full = full.reshape(-1)
    most = max((s[1] for s in shards), default=0)
    step = most if not chunk else int(chunk)
    for first in range(0, most, max(step, 1)):
        msgs, scat, staging = api.gather_plan_ex(shards, rank, height, width, first, max(step, 1), 0, root, shape)
        stage = torch.empty((staging,), dtype=local.dtype, device=local.device) if staging else None
        ops = []
        for peer, send, off, cnt in msgs:
            if send == 1:
                ops.append(dist.P2POp(dist.isend, flat_local[off:off + cnt], peer, group))
            elif send == 2:
                ops.append(dist.P2POp(dist.irecv, stage[off:off + cnt], peer, group))
            else:
                ops.append(dist.P2POp(dist.irecv, flat_full[off:off + cnt], peer, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for src, dst_off, run, n_runs, src_stride, dst_stride in scat:    # what slx_row_scatter_kernel does on the root
            for t in range(n_runs):
                flat_full[dst_off + t * dst_stride: dst_off + t * dst_stride + run] = stage[src + t * src_stride: src + t * src_stride + run]
    return full


def gather_rows(local, heights, dst=0, group=None, shape="in_place", chunk=None):
    """Row-tile gather: local [n_sets, h_rank, W]; heights = tile height of every rank -> [n_sets, sum(heights), W]."""
    n_sets, _, width = local.shape
    shards, row0 = [], 0
    for h in heights:
        shards.append((0, n_sets, row0, h))
        row0 += h
    return gather_shards(local, shards, row0, width, dst=dst, group=group, shape=shape, chunk=chunk)
