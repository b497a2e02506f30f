"""`python bench.py --gpus N` with no launcher in the environment starts its N ranks itself.  Here, without a GPU, the ranks
run the launcher self-test (gloo, known arrays, no decode): the parent must relay rank 0's one JSON line, report the world
size the ranks saw, and turn a failing rank into a non-zero exit."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"] + list(extra),
                          capture_output=True, text=True, timeout=300, env=env)


def test_parent_starts_two_ranks_and_relays_the_line():
    r = _run("--selftest-launcher")
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                                    # exactly one line on stdout
    d = json.loads(lines[0])
    assert d["metric"] == "launcher_selftest" and d["n_gpus"] == 2 and d["rccl_world_size"] == 2
    assert d["max_rank"] == 1.0                                         # MAX over ranks of the rank number
    # the key layout of the N > 1 line: the headline IS the row-tile split end to end (gather included), --steps steps of it
    rows = d["with_gather"]["rows"]
    assert d["value"] == rows["end_to_end"]["value"] > 0 and rows["steps"] == d["steps"] == 3
    assert d["ms_per_step"] == rows["end_to_end"]["ms_per_step"] and "kernel_only" in d
    assert all(d["with_gather"][k]["gathered_equals_local_decodes"] is True for k in ("rows", "framesets"))


def test_failing_rank_gives_nonzero_exit():
    r = _run("--selftest-launcher", "fail")
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_more_ranks_than_gpus_is_refused_without_the_rehearsal_flag():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this host has the GPUs the command asks for")
    r = _run()                                                          # no GPU in this container (and one on the GPU box)
    assert r.returncode != 0 and "refusing" in r.stderr
