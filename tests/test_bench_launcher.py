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
    assert d["metric"] == "launcher_selftest" and d["n_gpus"] == 2 and d["torch_world_size"] == 2 and d["rccl_world_size"] is None   # gloo: no RCCL communicator exists
    assert d["max_rank"] == 1.0                                         # MAX over ranks of the rank number
    # the key layout of the N > 1 line: the headline IS the row-tile split end to end (gather included), --steps steps of it
    # ... in the faster of the two gather shapes, both timed for --steps steps and both checked against the local arrays
    assert d["with_gather"]["value_is"] in ("rows", "rows_staged")
    rows = d["with_gather"][d["with_gather"]["value_is"]]
    assert d["config"]["gather_shape"] == rows["gather_shape"] and {d["with_gather"][k]["gather_shape"] for k in ("rows", "rows_staged")} == {"in_place", "staged"}
    assert d["value"] == rows["end_to_end"]["value"] > 0 and rows["steps"] == d["steps"] == 3
    assert d["value"] == max(d["with_gather"][k]["end_to_end"]["value"] for k in ("rows", "rows_staged"))
    assert d["ms_per_step"] == rows["end_to_end"]["ms_per_step"] and "kernel_only" in d
    assert all(d["with_gather"][k]["gathered_equals_local_decodes"] is True for k in ("rows", "rows_staged", "framesets"))


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


def test_traffic_probe_parses_the_counter_files(tmp_path, monkeypatch):
    """bench.py measures roofline.traffic by running itself under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`: ONE child run per
    counter launches every workload (the headline and the other_configs entries) a fixed number of times.  Here a stand-in profiler
    (a shell script that writes the counter files rocprofv3 would) checks the arithmetic around it: the decode kernels' dispatches
    only, from every counter file, cut into the workloads by dispatch order; FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-byte reads at
    64 bytes), WRITE_SIZE x 1024; and the fallbacks when the profiler fails or the dispatch count is not the launched one."""
    import argparse
    import stat
    import importlib.util
    spec = importlib.util.spec_from_file_location("slx_bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n1, n2 = bench.PROBE_LAUNCHES, bench.ROTATE
    fake = tmp_path / "rocprofv3"
    fake.write_text("""#!/bin/bash
# stand-in for rocprofv3: --pmc <COUNTER> ... -d <dir> -- <program>
while [ $# -gt 0 ]; do case "$1" in --pmc) C=$2; shift;; -d) D=$2; shift;; --) break;; esac; shift; done
[ -n "$FAKE_FAIL" ] && exit 3
echo "$@" >> $FAKE_ARGV_LOG
mkdir -p $D/box
V=432000; [ $C = WRITE_SIZE ] && V=576000
# one file per process, as the profiler writes them: the first holds no decode dispatch at all
{ echo '"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"'
  echo "7,\\"a_kernel_of_the_runtime\\",\\"$C\\",123"; } > $D/box/0_counter_collection.csv
# workload 1: N1 dispatches of the stream kernel (ids 100...), alternating V and V + 2000, in two files; a foreign kernel in between
{ echo '"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"'
  for i in $(seq 0 2 $((N1 - 1))); do echo "$((100 + i)),\\"void (anonymous namespace)::slx_stream_kernel<3>(SlxKParams)\\",\\"$C\\",$V"; done
  echo "103,\\"some_other_kernel\\",\\"$C\\",999999999"; } > $D/box/1_counter_collection.csv
{ echo '"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"'
  for i in $(seq 1 2 $((N1 - 1))); do echo "$((100 + i)),\\"void (anonymous namespace)::slx_stream_kernel<3>(SlxKParams)\\",\\"$C\\",$((V + 2000))"; done
  # workload 2: N2 dispatches of a strip kernel (ids 500...), a tenth of the bytes
  for i in $(seq 0 $((N2 - 1 - DROP))); do echo "$((500 + i)),\\"void (anonymous namespace)::slx_strip_kernel<3, 3, 0, 4, false>(SlxKParams)\\",\\"$C\\",$((V / 10))"; done
} > $D/box/2_counter_collection.csv
""".replace("\\\\", "\\"))
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("N1", str(n1))
    monkeypatch.setenv("N2", str(n2))
    monkeypatch.setenv("DROP", "0")
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")):
            monkeypatch.delenv(k)
    argv_log = tmp_path / "argv.log"
    monkeypatch.setenv("FAKE_ARGV_LOG", str(argv_log))
    args = argparse.Namespace(config="C4", sets_per_gpu=32, variant=0, tune=["strip_rows=8", "gray_plain=1"])
    workloads = [("headline", "C4", 32, (), 1), ("C4x1", "C4", 1, (), bench.ROTATE)]
    got = bench.traffic_probe(args, workloads)
    total, source = got["headline"]
    assert total == (433000.0 * 1024 * 2) + (577000.0 * 1024), source         # dispatches from every counter file, the decode kernel's only
    assert "measured in this run" in source and "%d / %d dispatches" % (n1, n1) in source
    total1, source1 = got["C4x1"]
    assert total1 == (43200.0 * 1024 * 2) + (57600.0 * 1024) and "%d / %d dispatches" % (n2, n2) in source1
    # the child runs the launch plan the timed region runs: every --tune item is forwarded, and so is the list of workloads
    lines = argv_log.read_text().splitlines()
    assert len(lines) == 2                                              # one child run per counter, whatever the number of workloads
    for ln in lines:
        assert "--tune strip_rows=8 --tune gray_plain=1" in ln and "--probe-list headline:C4:32::1;C4x1:C4:1::%d" % bench.ROTATE in ln, ln
    assert bench.parse_probe_list("a:C4:16:xyUk:1;b:REF:1::12") == [("a", "C4", 16, ("x", "y", "U", "k"), 1), ("b", "REF", 1, (), 12)]
    # a pass whose decode dispatches are not the launched number is not cut into workloads by guesswork
    monkeypatch.setenv("DROP", "1")
    got = bench.traffic_probe(args, workloads)
    assert all(v[0] is None for v in got.values()) and "decode dispatches" in got["headline"][1]
    monkeypatch.setenv("DROP", "0")
    monkeypatch.setenv("FAKE_FAIL", "1")
    total, source = bench.traffic_probe(args, workloads)["headline"]
    assert total is None and "failed" in source
    monkeypatch.delenv("FAKE_FAIL")
    monkeypatch.setenv("ROCPROFILER_FAKE", "1")
    total, source = bench.traffic_probe(args, workloads)["C4x1"]
    assert total is None and "already runs under a profiler" in source
