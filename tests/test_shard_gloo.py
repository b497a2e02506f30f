"""world_size-2 (and 3) CPU tests of the multi-GPU path: partition by frame-set and by
row-tile, one gather of the finished depth maps, result equal to the single-rank result.
The per-rank decode is played by the oracle here (there is no GPU in this container); on
the GPU box tests/test_gpu_parity.py::test_row_tiles_and_frameset_shards_on_gpu runs the
same partition through the HIP path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _small_spec(synth):
    spec = dict(synth.make_spec("C2"))
    spec["width"], spec["height"] = 48, 31          # 31 rows: no world size of the tests divides it
    spec["calib"] = synth.scaled_calibration(48, 31, spec["proj_width"])
    return spec


def _worker(rank, world, port, n_sets, by, outdir):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle as O
    synth, shard = pkg("synth"), pkg("shard")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = _small_spec(synth)
        sets = [synth.random_planes(spec, seed=100 + s)[0] for s in range(n_sets)]
        if by == "frameset":
            lo, hi = shard.split_range(n_sets, world, rank)
            local = [O.pipeline(spec, sets[s], None, want=("z",))["z"] for s in range(lo, hi)]
            local = torch.from_numpy(np.stack(local)) if local else torch.zeros((0, spec["height"], spec["width"]), dtype=torch.float64)
            full = shard.gather_depth(local, dst=0)
            if rank == 0:
                np.save(os.path.join(outdir, "frameset.npy"), full.numpy())
            everyone = shard.gather_depth(local, all_ranks=True)
            assert everyone.shape[0] == n_sets
            # the shard-table gather (same messages as the native RCCL one) delivers the same array, on every rank too
            table = shard.shards_by_frameset(n_sets, world, spec["height"])
            again = shard.gather_shards(local, table, spec["height"], spec["width"], dst=None)
            assert torch.equal(again, everyone)
        else:
            tile, lo, hi = shard.row_tile_spec(spec, world, rank)
            local = np.stack([O.pipeline(tile, sets[s][:, lo:hi], None, want=("z",))["z"] for s in range(n_sets)])
            # the library call as the bench uses it: [n_sets, rows of this rank, W] in, [n_sets, H, W] out, no reshuffling here
            heights = [shard.split_range(spec["height"], world, r)[1] - shard.split_range(spec["height"], world, r)[0] for r in range(world)]
            full = shard.gather_rows(torch.from_numpy(local), heights, dst=0)
            assert (full is None) == (rank != 0)
            if rank == 0:
                np.save(os.path.join(outdir, "rows.npy"), full.numpy())
            last = world - 1
            full_last = shard.gather_shards(torch.from_numpy(local), shard.shards_by_rows(n_sets, world, spec["height"]), spec["height"],
                                            spec["width"], dst=last)
            if rank == last:
                np.save(os.path.join(outdir, "rows_last.npy"), full_last.numpy())
            # the STAGED shape of the same gather (one message per (peer, chunk) into a staging buffer + the planner's row scatter),
            # in chunks of 2 frame-sets with a ragged last chunk, to rank 0 and to the last rank
            for dst_rank, fname in ((0, "rows_staged.npy"), (last, "rows_staged_last.npy")):
                staged = shard.gather_shards(torch.from_numpy(local), shard.shards_by_rows(n_sets, world, spec["height"]), spec["height"],
                                             spec["width"], dst=dst_rank, shape="staged", chunk=2)
                assert (staged is None) == (rank != dst_rank)
                if rank == dst_rank:
                    np.save(os.path.join(outdir, fname), staged.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_sets,by", [(2, 4, "frameset"), (2, 3, "frameset"), (3, 4, "rows"), (2, 2, "rows"), (4, 3, "rows"), (3, 2, "frameset")])
def test_sharded_decode_and_gather(tmp_path, world, n_sets, by):
    import oracle as O
    synth = pkg("synth")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_sets, by, str(tmp_path)), nprocs=world, join=True)
    spec = _small_spec(synth)
    want = np.stack([O.pipeline(spec, synth.random_planes(spec, seed=100 + s)[0], None, want=("z",))["z"] for s in range(n_sets)])
    got = np.load(os.path.join(str(tmp_path), "frameset.npy" if by == "frameset" else "rows.npy"))
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)
    if by == "rows":
        for fname in ("rows_last.npy", "rows_staged.npy", "rows_staged_last.npy"):
            assert np.array_equal(np.load(os.path.join(str(tmp_path), fname)), want, equal_nan=True), fname
