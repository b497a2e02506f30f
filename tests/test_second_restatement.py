"""tests/numpy_restatement.py (a second, independent reading of the reference, in numpy) against oracle/slx_oracle.c, bit for
bit: every possible input of CDecodePhase::CountResult, the committed scenes, unstructured bytes in every mode.  Two
implementations that agree are still not the reference: parity stays "unpinned" (DESIGN.md section 6); what this buys is
that a misreading would have to be made twice, in two languages and two program structures, to go unnoticed."""
import json
import os

import numpy as np
import pytest

import numpy_restatement as R2


def _exhaustive_planes():
    d = np.arange(-255, 256)
    d02, d13 = np.meshgrid(d, d, indexing="ij")
    p = np.zeros((4, 511, 511), dtype=np.uint8)
    p[0], p[2] = np.maximum(d02, 0), np.maximum(-d02, 0)
    p[1], p[3] = np.maximum(d13, 0), np.maximum(-d13, 0)
    return p


def _same(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("T", [40, 20, 30, 160, 240, 1280, 1920, 4096, 8, 64, 512, 7, 1000003])
def test_wrapped_phase_all_511x511_inputs(oracle, synth, T):
    planes = _exhaustive_planes()
    spec = {"width": 511, "height": 511, "mode": synth.MODE_PHASE_ONLY, "n_freq": 1, "n_steps": 4, "periods": [T]}
    a = oracle.pipeline(spec, planes, None, want=("pix",))["pix"][0]
    b = R2.wrapped_phase(planes, T)
    assert _same(a, b), int((a != b).sum())
    assert b.min() > 0.0 and b.max() <= T + 0.5


def test_fast_atan2_known_answers():
    f = lambda y, x: float(R2.fast_atan2_deg(np.float32(y), np.float32(x)))
    assert f(0, 1) == 0.0 and f(0, 0) == 0.0
    assert abs(f(1, 0) - 90.0) < 1e-3 and abs(f(0, -1) - 180.0) < 1e-3 and abs(f(-1, 0) - 270.0) < 1e-3
    y, x = np.meshgrid(np.arange(-255, 256, dtype=np.float32) / 2, np.arange(-255, 256, dtype=np.float32) / 2, indexing="ij")
    a = R2.fast_atan2_deg(y, x).astype(np.float64)
    t = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360.0
    err = np.abs((a - t + 180.0) % 360.0 - 180.0)
    assert err.max() <= 0.3 and a.min() >= 0.0 and a.max() <= 360.0      # the accuracy OpenCV documents for fastAtan2


@pytest.mark.parametrize("name", ["C1x4", "C2", "C3", "C5"])
def test_committed_scenes(synth, golden_dir, name):
    d = np.load(os.path.join(golden_dir, "scene_%s.npz" % name))
    spec = dict(synth.make_spec(name))
    spec["width"], spec["height"] = 64, 48
    spec["calib"] = synth.scaled_calibration(64, 48, spec["proj_width"])
    got = R2.pipeline(spec, d["phase"] if "phase" in d.files else None, d["gray_planes"] if "gray_planes" in d.files else None)
    checked = 0
    for k in d.files:
        if k.startswith("out_"):
            assert _same(got[k[4:]], d[k]), k
            checked += 1
    assert checked >= 6


@pytest.mark.parametrize("name", ["C1", "C1x4", "C2", "C3", "C4"])
@pytest.mark.parametrize("shape", [(37, 130), (5, 17), (1, 1)])
def test_unstructured_bytes_every_mode(oracle, synth, name, shape):
    h, w = shape
    spec = dict(synth.make_spec(name))
    spec["width"], spec["height"] = w, h
    spec["calib"] = synth.scaled_calibration(w, h, spec["proj_width"])
    ph, gr = synth.random_planes(spec, seed=h * 131 + w)
    if gr is not None and w > 8:
        gr[:, :, : w // 2] = np.where(gr[:, :, : w // 2] > 127, 220, 20)
        gr[1::2, :, : w // 4] = gr[0::2, :, : w // 4]                 # exact ties
    want = ["z", "x", "y", "U", "pix"]
    if spec.get("gray_bits"):
        want.append("gray")
    if spec["mode"] in (3, 4):
        want.append("mask")
        if spec["n_freq"] > 1:
            want.append("k")
    a = oracle.pipeline(spec, ph, gr, want=tuple(want))
    b = R2.pipeline(spec, ph, gr)
    for k in want:
        assert _same(a[k], b[k]), (k, int((a[k] != b[k]).sum()))


@pytest.mark.parametrize("n_steps", [3, 5, 8, 16])
def test_n_step_sums(oracle, synth, n_steps):
    spec = {"width": 97, "height": 33, "mode": synth.MODE_PHASE_ONLY, "n_freq": 1, "n_steps": n_steps, "periods": [512]}
    ph, _ = synth.random_planes(spec, seed=n_steps)
    a = oracle.pipeline(spec, ph, None, want=("pix",))["pix"][0]
    assert _same(a, R2.wrapped_phase_nstep(list(ph), 512))


def test_gray_table_file_and_arbitrary_table(oracle, synth, golden_dir):
    rows = json.load(open(os.path.join(golden_dir, "vGrayCode_rows.json")))["rows"]
    t = R2.gray_table(rows, 64)
    assert np.array_equal(t, synth.standard_gray_lut(6))             # the reference's own table
    spec = {"width": 64, "height": 31, "mode": synth.MODE_GRAY_ONLY, "gray_bits": 6, "gray_stripe": 20, "gray_lut": t}
    _, gr = synth.random_planes(spec, seed=3)
    assert _same(oracle.pipeline(spec, None, gr, want=("gray",))["gray"], R2.gray_decode(gr, t, 20)[0])
    odd = (np.arange(64)[::-1] - 10).astype(np.int16)                # negative entries: the merge's parity test on negative stripes
    spec2 = dict(synth.make_spec("C1x4"), width=64, height=31, gray_lut=odd)
    spec2["calib"] = synth.scaled_calibration(64, 31, spec2["proj_width"])
    ph, gr = synth.random_planes(spec2, seed=4)
    a = oracle.pipeline(spec2, ph, gr, want=("U", "z"))
    b = R2.pipeline(spec2, ph, gr)
    assert _same(a["U"], b["U"]) and _same(a["z"], b["z"])
