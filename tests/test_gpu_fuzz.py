"""A few seconds of each differential fuzzer (tools/fuzz_parity.py, fuzz_track.py, fuzz_api.py) with fixed seeds, as part of the GPU suite:
random shapes, modes, strides, optional planes, launch plans, call sequences, tracker feeds and calibrations against the oracle.
The long runs are recorded in profiles/r04_fuzz_parity.log; these keep the fuzzers themselves alive and catch a regression that
the pinned geometries of test_gpu_parity.py would step over."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, seconds, seed, profile=None):
    env = dict(os.environ)
    if profile:
        env["FUZZ_PROFILE"] = profile
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(seconds), str(seed)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    return r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("profile,seed", [(None, 11), ("strip", 12), ("big", 13), ("bigstrip", 14), ("calib", 15)])
def test_decode_fuzz(profile, seed):
    out = _run("fuzz_parity.py", 6, seed, profile)
    assert " 0 failures" in out and "fuzz_parity: 0 cases" not in out, out[-2000:]


@pytest.mark.gpu
def test_tracker_and_cloud_fuzz():
    out = _run("fuzz_track.py", 8, 21)
    assert " 0 failures" in out, out[-2000:]


@pytest.mark.gpu
def test_call_sequence_fuzz():
    out = _run("fuzz_api.py", 8, 31)
    assert " 0 failures" in out, out[-2000:]
