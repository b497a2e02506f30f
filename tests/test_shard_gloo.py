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
    spec["width"], spec["height"] = 48, 30
    spec["calib"] = synth.scaled_calibration(48, 30, spec["proj_width"])
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
        else:
            tile, lo, hi = shard.row_tile_spec(spec, world, rank)
            local = np.stack([O.pipeline(tile, sets[s][:, lo:hi], None, want=("z",))["z"] for s in range(n_sets)])
            # gather along rows: move the row axis first
            t = torch.from_numpy(np.ascontiguousarray(local.transpose(1, 0, 2)))
            full = shard.gather_depth(t, dst=0)
            if rank == 0:
                np.save(os.path.join(outdir, "rows.npy"), full.numpy().transpose(1, 0, 2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_sets,by", [(2, 4, "frameset"), (2, 3, "frameset"), (3, 4, "rows"), (2, 2, "rows")])
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
