"""A fixed number of cases of each differential fuzzer (tools/fuzz_parity.py, fuzz_track.py, fuzz_api.py) with fixed seeds, as part of
the GPU suite: random shapes, modes, strides, optional planes, launch plans, call sequences, tracker feeds and calibrations against
the oracle.  Every leg runs `--cases N`: the case list is a function of (profile, seed, N) alone -- not of the box's speed or of the
oracle's CPU time -- so two GPUTEST records ran the same cases; the count each leg reports is asserted.  The long runs are recorded
under profiles/ (r04_fuzz_parity.log, r05_fuzz_parity.log, r06_fuzz_parity.log); these keep the fuzzers themselves alive and catch a regression that the
pinned geometries of test_gpu_parity.py would step over."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, cases, seed, profile=None):
    env = dict(os.environ)
    env.pop("FUZZ_PROFILE", None)
    if profile:
        env["FUZZ_PROFILE"] = profile
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), "0", str(seed), "--cases", str(cases)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    print(r.stdout[-1500:])                                           # the case classes, in the pytest log (-s / on failure)
    return r.stdout


def _count(out, what):
    m = re.search(r"(\d+) %s" % what, out)
    assert m, out[-2000:]
    return int(m.group(1))


@pytest.mark.gpu
@pytest.mark.parametrize("profile,seed,cases", [(None, 11, 800), ("strip", 12, 80), ("big", 13, 250), ("bigstrip", 14, 350), ("calib", 15, 300), ("refstream", 16, 120)])
def test_decode_fuzz(profile, seed, cases):
    out = _run("fuzz_parity.py", cases, seed, profile)
    assert " 0 failures" in out, out[-2000:]
    assert _count(out, "cases, ") == cases, out[-2000:]


@pytest.mark.gpu
def test_tracker_and_cloud_fuzz():
    out = _run("fuzz_track.py", 300, 21)
    assert " 0 failures" in out, out[-2000:]
    assert _count(out, r"cases \(") == 300, out[-2000:]


@pytest.mark.gpu
def test_call_sequence_fuzz():
    out = _run("fuzz_api.py", 500, 31)
    assert " 0 failures" in out, out[-2000:]
    assert _count(out, "contexts, ") == 500, out[-2000:]
