"""The native gather's message schedule (csrc/slx_comm.cpp: plan_range, the code gather_range posts from), checked without
a GPU for worlds of 2..8 ranks: every rank's plan is played against the others' the way RCCL matches grouped sends and
receives -- per ordered pair of ranks, in posting order, lengths must agree -- and the replayed copies must reassemble
[set][H][W] exactly, for both splits, ragged tiles, empty shards, chunked gathers, a root other than 0 and all-ranks gathers.
The 1-GPU boxes cannot host an N > 1 RCCL run; this is the evidence that the schedule cannot deadlock or misplace a tile."""
import numpy as np
import pytest


def play(api, shards, H, W, chunk, root, in_place_on_receivers=True):
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
        plans = {r: api.gather_plan(shards, r, H, W, first, chunk, local_plane_stride=(lstride[r] if shards[r][3] != H or lstride[r] == H * W else 0), root=root)
                 for r in range(world)}
        sends = {(r, d): [] for r in range(world) for d in range(world)}
        recvs = {(r, d): [] for r in range(world) for d in range(world)}
        for r, plan in plans.items():
            for peer, send, off, cnt in plan:
                assert peer != r and 0 <= peer < world
                (sends[(r, peer)] if send else recvs[(peer, r)]).append((off, cnt))
        for pair in sends:
            a, b = sends[pair], recvs[pair]
            assert len(a) == len(b), ("unmatched messages", pair, len(a), len(b))        # else a rank waits forever
            for (so, sc), (ro, rc) in zip(a, b):
                assert sc == rc, ("length mismatch", pair, sc, rc)
                src, dst = pair
                full[dst][ro:ro + rc] = local[src][lbase[src] + so:lbase[src] + so + sc]
                n_msgs += 1
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
