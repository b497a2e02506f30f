"""The native gather's message schedule (csrc/slx_comm.cpp: plan_range, the code gather_range posts from), checked without
a GPU for worlds of 2..8 ranks: every rank's plan is played against the others' the way RCCL matches grouped sends and
receives -- per ordered pair of ranks, in posting order, lengths must agree -- and the replayed copies must reassemble
[set][H][W] exactly, for both splits, ragged tiles, empty shards, chunked gathers, a root other than 0 and all-ranks gathers.
The 1-GPU boxes cannot host an N > 1 RCCL run; this is the evidence that the schedule cannot deadlock or misplace a tile."""
import numpy as np
import pytest


def play(api, shards, H, W, chunk, root, in_place_on_receivers=True, shape="in_place"):
    world = len(shards)
    total = max(s0 + n for s0, n, _, _ in shards)
    truth = np.arange(total * H * W, dtype=np.float64).reshape(total, H, W) + 0.25
    receivers = [r for r in range(world) if root < 0 or root == r]
    # every rank's local buffer: in place inside `full` on receivers (plane stride H*W), a dense tile stack elsewhere
    full = {r: np.full(total * H * W, -1.0) for r in receivers}
    local, lstride, lbase = {}, {}, {}
    for r, (s0, n, r0, rows) in enumerate(shards):
        if r in receivers and in_place_on_receivers:
            f = full[r].reshape(total, H, W)
            f[s0:s0 + n, r0:r0 + rows] = truth[s0:s0 + n, r0:r0 + rows]
            local[r], lstride[r], lbase[r] = full[r], H * W, (s0 * H + r0) * W
        else:
            local[r] = np.ascontiguousarray(truth[s0:s0 + n, r0:r0 + rows]).reshape(-1)
            lstride[r], lbase[r] = rows * W, 0
            if r in receivers:
                full[r].reshape(total, H, W)[s0:s0 + n, r0:r0 + rows] = truth[s0:s0 + n, r0:r0 + rows]   # the self copy gather_range does
    most = max(n for _, n, _, _ in shards)
    n_msgs = 0
    for first in range(0, max(most, 1), chunk):
        plans, scatters, stage = {}, {}, {}
        for r in range(world):
            ls = lstride[r] if shards[r][3] != H or lstride[r] == H * W else 0
            plans[r], scatters[r], staging = api.gather_plan_ex(shards, r, H, W, first, chunk, local_plane_stride=ls, root=root, shape=shape)
            stage[r] = np.full(staging, -2.0)
            if shape == "in_place":
                assert not scatters[r] and staging == 0
                assert plans[r] == api.gather_plan(shards, r, H, W, first, chunk, local_plane_stride=ls, root=root)      # the two entry points agree
        sends = {(r, d): [] for r in range(world) for d in range(world)}
        recvs = {(r, d): [] for r in range(world) for d in range(world)}
        for r, plan in plans.items():
            for peer, send, off, cnt in plan:
                assert peer != r and 0 <= peer < world and send in (0, 1, 2)
                (sends[(r, peer)] if send == 1 else recvs[(peer, r)]).append((off, cnt, send))
        for pair in sends:
            a, b = sends[pair], recvs[pair]
            assert len(a) == len(b), ("unmatched messages", pair, len(a), len(b))        # else a rank waits forever
            for (so, sc, _), (ro, rc, kind) in zip(a, b):
                assert sc == rc, ("length mismatch", pair, sc, rc)
                src, dst = pair
                target = stage[dst] if kind == 2 else full[dst]
                assert ro + rc <= target.size
                target[ro:ro + rc] = local[src][lbase[src] + so:lbase[src] + so + sc]
                n_msgs += 1
        for r in range(world):                                                          # the root's row scatter, as slx_row_scatter_kernel does it
            covered = np.zeros(stage[r].size, dtype=bool)
            for src, dst, run, n_runs, src_stride, dst_stride in scatters[r]:
                for t in range(n_runs):
                    full[r][dst + t * dst_stride: dst + t * dst_stride + run] = stage[r][src + t * src_stride: src + t * src_stride + run]
                    assert not covered[src + t * src_stride: src + t * src_stride + run].any()
                    covered[src + t * src_stride: src + t * src_stride + run] = True
            assert covered.all()                                                        # every staged double goes somewhere, once
    for r in receivers:
        assert np.array_equal(full[r].reshape(total, H, W), truth), r
    return n_msgs


@pytest.mark.parametrize("world", [2, 3, 4, 5, 8])
@pytest.mark.parametrize("split", ["rows", "framesets"])
@pytest.mark.parametrize("root", [0, -1, "last"])
def test_schedule_reassembles_for_every_world(api, shard, world, split, root):
    H, W, total = 37, 6, 11                                            # 37 rows, 11 sets: nothing divides
    root = world - 1 if root == "last" else root
    shards = (shard.shards_by_rows if split == "rows" else shard.shards_by_frameset)(total, world, H)
    for chunk in (1, 4, 100):
        n = play(api, shards, H, W, chunk, root)
        n_recv = world if root < 0 else 1
        if split == "rows":
            assert n == n_recv * (world - 1) * total                   # one message per (receiver, peer, frame-set)
    play(api, shards, H, W, 3, root, in_place_on_receivers=False)


@pytest.mark.parametrize("world", [2, 3, 4, 5, 8])
@pytest.mark.parametrize("split", ["rows", "framesets"])
@pytest.mark.parametrize("root", [0, -1, "last"])
def test_staged_schedule_reassembles_for_every_world(api, shard, world, split, root):
    """The STAGED shape: one message per (peer, chunk) into the root's staging slot, then the planner's scatter list.  Same truth
    array, same matching rules; with one root a row split posts (world - 1) messages per chunk instead of (world - 1) x sets;
    an all-ranks gather (root = -1) and whole-frame shards keep the in-place messages."""
    H, W, total = 37, 6, 11
    root = world - 1 if root == "last" else root
    shards = (shard.shards_by_rows if split == "rows" else shard.shards_by_frameset)(total, world, H)
    for chunk in (1, 4, 100):
        n = play(api, shards, H, W, chunk, root, shape="staged")
        n_chunks = (total + chunk - 1) // chunk
        if split == "rows" and root >= 0:
            assert n == (world - 1) * n_chunks                          # one message per (peer, chunk)
        elif split == "rows":
            assert n == world * (world - 1) * total                     # every rank receives in place: the in-place shape
    play(api, shards, H, W, 3, root, in_place_on_receivers=False, shape="staged")


def test_staged_schedule_needs_a_dense_tile_stack_on_senders(api, shard):
    rows = shard.shards_by_rows(4, 3, 30)
    with pytest.raises(api.SlxError):
        api.gather_plan_ex(rows, 1, 30, 8, 0, 4, local_plane_stride=30 * 8, root=0, shape="staged")      # rank 1 sends from a full-height layout
    msgs, scat, staging = api.gather_plan_ex(rows, 1, 30, 8, 0, 4, local_plane_stride=0, root=0, shape="staged")
    assert msgs == [(0, 1, 0, 4 * 10 * 8)] and not scat and staging == 0
    with pytest.raises(api.SlxError):
        api.gather_plan_ex(rows, 0, 30, 8, 0, 4, root=0, shape=7)


def test_schedule_with_empty_shards_and_more_ranks_than_work(api, shard):
    assert play(api, shard.shards_by_frameset(3, 8, 20), 20, 4, 2, 0) == 2      # 5 of 8 ranks hold nothing
    assert play(api, shard.shards_by_rows(2, 8, 5), 5, 4, 1, -1) > 0            # 3 of 8 row tiles are empty
    with pytest.raises(api.SlxError):
        api.gather_plan([(0, 2, 0, 10), (2, 2, 0, 10)], 0, 10, 4, 0, 4, local_plane_stride=50, root=1)   # whole-frame shard, not dense


def test_config4_plan_sizes(api, shard):
    """BASELINE configuration 4: 256 frame-sets of 1920 x 1200 over 8 ranks, chunks of 8 frame-sets."""
    H, W = 1200, 1920
    rows = shard.shards_by_rows(256, 8, H)
    plan0 = api.gather_plan(rows, 0, H, W, 0, 8, local_plane_stride=H * W, root=0)
    assert len(plan0) == 7 * 8 and all(not send and cnt == 150 * W for _, send, _, cnt in plan0)
    plan3 = api.gather_plan(rows, 3, H, W, 8, 8, root=0)
    assert len(plan3) == 8 and all(send and peer == 0 and cnt == 150 * W for peer, send, _, cnt in plan3)
    sets = shard.shards_by_frameset(256, 8, H)
    assert api.gather_plan(sets, 0, H, W, 0, 8, local_plane_stride=H * W, root=0) == [(p, 0, (32 * p) * H * W, 8 * H * W) for p in range(1, 8)]
    # the staged shape of the same step: 7 messages of 8 x 150 x 1920 doubles (18.4 MB) per chunk into a 129 MB slot, 7 scatter segments
    msgs, scat, staging = api.gather_plan_ex(rows, 0, H, W, 0, 8, local_plane_stride=H * W, root=0, shape="staged")
    assert [m[:2] for m in msgs] == [(p, 2) for p in range(1, 8)] and all(m[3] == 8 * 150 * W for m in msgs)
    assert staging == 7 * 8 * 150 * W and len(scat) == 7
    assert scat[2] == (2 * 8 * 150 * W, 3 * 150 * W, 150 * W, 8, 150 * W, H * W)
    msgs3, scat3, _ = api.gather_plan_ex(rows, 3, H, W, 8, 8, root=0, shape="staged")
    assert msgs3 == [(0, 1, 8 * 150 * W, 8 * 150 * W)] and not scat3
