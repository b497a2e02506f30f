"""CPU tests of the oracle (oracle/, test infrastructure) against the two data files the
reference holds (vGrayCode.txt, Result.yml), against known answers of the algorithms it
restates, and against the committed oracle-generated regression vectors.

The reference ships no tests or golden outputs and is unbuildable here (OpenCV 2.4.9 is
absent), so apart from the two data files these checks do not pin the oracle to the
reference: "parity unpinned".
"""
import hashlib
import json
import math
import os

import numpy as np
import pytest

GOLDEN_SCENES = ["C1x4", "C2", "C3", "C5"]


def small_spec(synth, name, w=64, h=48):
    spec = dict(synth.make_spec(name))
    spec["width"], spec["height"] = w, h
    spec["calib"] = synth.scaled_calibration(w, h, spec["proj_width"])
    return spec


def exhaustive_planes():
    d = np.arange(-255, 256)
    d02, d13 = np.meshgrid(d, d, indexing="ij")
    p = np.zeros((4, 511, 511), dtype=np.uint8)
    p[0] = np.maximum(d02, 0)
    p[2] = np.maximum(-d02, 0)
    p[1] = np.maximum(d13, 0)
    p[3] = np.maximum(-d13, 0)
    return p


# ---------------------------------------------------------------- reference data files
def test_gray_table_matches_reference_file(oracle, synth, golden_dir):
    rows = json.load(open(os.path.join(golden_dir, "vGrayCode_rows.json")))["rows"]
    assert len(rows) == 64
    lut = oracle.gray_lut_from_rows(rows)            # R/CDecodeGray.cpp:120-125
    for b, g in rows:
        assert g == b ^ (b >> 1)                     # the file is the reflected Gray code
        assert lut[g] == b
    assert np.array_equal(lut, synth.standard_gray_lut(6))


def test_calibration_matches_reference_file(oracle, synth, golden_dir):
    yml = json.load(open(os.path.join(golden_dir, "result_yml.json")))
    assert synth.RESULT_YML["cam"] == yml["CamMat"]
    assert synth.RESULT_YML["pro"] == yml["ProMat"]
    assert synth.RESULT_YML["rot"] == yml["R"]
    assert synth.RESULT_YML["trans"] == yml["T"]
    P = oracle.projection_matrix(yml["ProMat"], yml["R"], yml["T"])
    Kp = np.array(yml["ProMat"]).reshape(3, 3)
    RT = np.hstack([np.array(yml["R"]).reshape(3, 3), np.array(yml["T"]).reshape(3, 1)])
    assert np.allclose(P, Kp @ RT, rtol=1e-14, atol=0)
    # R is a rotation: the file is self-consistent
    R = np.array(yml["R"]).reshape(3, 3)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-9)


# ---------------------------------------------------------------- cvFastArctan (a2)
def test_fast_atan_known_answers(oracle):
    f = oracle.fast_atan2_deg
    assert f(0.0, 5.0) == 0.0
    assert f(0.0, 0.0) == 0.0
    assert abs(f(5.0, 0.0) - 90.0) < 1e-4
    assert abs(f(0.0, -5.0) - 180.0) < 1e-4
    assert abs(f(-5.0, 0.0) - 270.0) < 1e-4
    assert abs(f(1.0, 1.0) - 45.0) < 0.3


def test_fast_atan_error_bound(oracle):
    rng = np.random.default_rng(1)
    worst = 0.0
    for _ in range(4000):
        y, x = rng.normal(size=2) * 100
        a = oracle.fast_atan2_deg(y, x)
        assert 0.0 <= a <= 360.0
        t = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - t)
        worst = max(worst, min(d, 360.0 - d))
    assert worst <= 0.3, worst                       # OpenCV documents ~0.3 degrees


# ---------------------------------------------------------------- wrapped phase (a1)
def test_wrapped_phase_exhaustive_tables(oracle, synth, golden_dir):
    tables = json.load(open(os.path.join(golden_dir, "wrapped_phase_tables.json")))["tables"]
    planes = exhaustive_planes()
    for T, entry in tables.items():
        T = int(T)
        spec = {"width": 511, "height": 511, "mode": synth.MODE_PHASE_ONLY, "n_freq": 1, "n_steps": 4, "periods": [T]}
        pix = oracle.pipeline(spec, planes, None, want=("pix",))["pix"][0]
        assert hashlib.sha256(np.ascontiguousarray(pix).tobytes()).hexdigest() == entry["sha256"], T
        for i, j, v in entry["samples"]:
            assert pix[i, j] == v
        assert pix.min() > 0.0 and pix.max() <= T + 0.5           # (0, T+0.5]; 0.5 at zero phase
        assert np.all(pix == pix.astype(np.float32))              # float-valued, stored as double
        assert pix[255, 255] == 0.5                               # all four images equal -> x = 0


def test_wrapped_phase_recovers_rendered_phase(oracle, synth):
    spec = small_spec(synth, "C1x4")
    T = spec["periods"][0]
    ph, gr, U = synth.render(spec, "tilted")
    pix = oracle.pipeline(dict(spec, mode=synth.MODE_PHASE_ONLY), ph, None, want=("pix",))["pix"][0]
    lit = (U >= 0) & (U < spec["proj_width"])
    d = (pix - 0.5 - np.fmod(U, T) + T / 2) % T - T / 2          # wrapped difference
    assert np.abs(d[lit]).max() < 0.3 / 360 * T + 0.25           # atan error + 8-bit quantisation


def test_generic_nstep_reduces_to_4step(oracle, synth):
    rng = np.random.default_rng(2)
    planes = rng.integers(0, 256, size=(4, 33, 47), dtype=np.uint8)
    spec = {"width": 47, "height": 33, "mode": synth.MODE_PHASE_ONLY, "n_freq": 1, "n_steps": 4, "periods": [40]}
    a1 = oracle.pipeline(spec, planes, None, want=("pix",))["pix"][0]
    x1 = oracle.wrapped_phase_generic(planes, 40)
    assert np.array_equal(a1, x1)


def test_nstep_weights(oracle):
    wy, wx, sc = oracle.nstep_weights(4)
    assert wy.tolist() == [1.0, 0.0, -1.0, 0.0] and wx.tolist() == [0.0, 1.0, 0.0, -1.0] and sc == 0.5
    wy, wx, sc = oracle.nstep_weights(8)
    r = np.float32(math.sqrt(0.5))
    assert sc == 0.25
    assert np.allclose(wy, [1, r, 0, -r, -1, -r, 0, r], atol=1e-7)
    assert wy[2] == 0.0 and wx[0] == 0.0 and wx[4] == 0.0


def test_nstep_recovers_phase(oracle, synth):
    spec = small_spec(synth, "C5")
    ph, _, U = synth.render(spec, "tilted")
    res = oracle.pipeline(spec, ph, None, want=("U", "k"))
    assert np.abs(res["U"] - 0.5 - U).max() < 0.2


# ---------------------------------------------------------------- Gray decode, merge (a3-a5)
def test_gray_decode_and_merge(oracle, synth):
    spec = small_spec(synth, "C1x4")
    S, T = spec["gray_stripe"], spec["periods"][0]
    assert (S, T) == (20, 40)                                     # R/CCalculation.cpp:550, :562-563
    ph, gr, U = synth.render(spec, "tilted")
    res = oracle.pipeline(spec, ph, gr, want=("gray", "U", "z"))
    lit = (U >= 0) & (U < spec["proj_width"])
    assert np.array_equal(res["gray"][lit], (np.floor(U / S) * S)[lit])
    assert np.abs(res["U"] - 0.5 - U)[lit].max() < 0.3
    z_true = synth.scene_depth(spec, "tilted")
    ok = res["z"] > 0
    assert ok.mean() > 0.95
    assert np.abs(res["z"] - z_true)[ok].max() < 5.0              # the +0.5 px bias of a1 is intended


def test_gray_threshold_ties_are_zero(oracle, synth):
    spec = small_spec(synth, "C1x4", 8, 2)
    spec["mode"] = synth.MODE_GRAY_ONLY
    gr = np.full((12, 2, 8), 100, dtype=np.uint8)                 # pattern == inverse -> bit 0
    res = oracle.pipeline(spec, None, gr, want=("gray",))
    assert np.all(res["gray"] == 0)
    gr[0] = 101                                                   # bit 0 (LSB) set -> gray 1 -> bin 1
    res = oracle.pipeline(spec, None, gr, want=("gray",))
    assert np.all(res["gray"] == spec["gray_stripe"])
    gr[0] = 100
    gr[10] = 200                                                  # bit 5 -> gray 32 -> bin 63
    res = oracle.pipeline(spec, None, gr, want=("gray",))
    assert np.all(res["gray"] == 63 * spec["gray_stripe"])


# ---------------------------------------------------------------- unwrap and mask (x2, x3)
def test_unwrap_consistency(oracle, synth):
    spec = small_spec(synth, "C2")
    ph, _, U = synth.render(spec, "sphere", noise_sigma=2.0)
    res = oracle.pipeline(spec, ph, None, want=("U", "pix", "k"))
    T = spec["periods"]
    Uf = res["pix"][0].copy()
    for f in (1, 2):
        k = np.floor((Uf - res["pix"][f]) / T[f] + 0.5).astype(np.int32)
        assert np.array_equal(k, res["k"][f - 1])
        Uf = res["pix"][f] + k * T[f]
    assert np.array_equal(Uf, res["U"])
    lit = (U >= 8) & (U < spec["proj_width"] - 8)               # the unit-frequency phase wraps at the projector's edges
    assert np.abs(res["U"] - 0.5 - U)[lit].max() < 0.5


def test_gray_mask_three_tap(oracle, synth):
    spec = small_spec(synth, "C3")
    ph, gr, U = synth.render(spec, "tilted")
    res = oracle.pipeline(spec, ph, gr, want=("mask", "z"))
    assert res["mask"].all()
    bad = gr.copy()
    r, c = 10, 20
    for b in range(6):                                            # wreck one pixel's Gray word
        bad[2 * b, r, c], bad[2 * b + 1, r, c] = bad[2 * b + 1, r, c], bad[2 * b, r, c]
    res2 = oracle.pipeline(spec, ph, bad, want=("mask", "z"))
    inv = np.argwhere(res2["mask"] == 0)
    assert sorted(map(tuple, inv)) == [(r, c - 1), (r, c), (r, c + 1)]
    assert np.all(res2["z"][res2["mask"] == 0] == 0)
    keep = res2["mask"] == 1
    assert np.array_equal(res2["z"][keep], res["z"][keep])


def test_point_cloud_order_and_filter(oracle, synth):
    """CCalculation::Result: column outer / row inner, only depths inside [fov_min, fov_max]."""
    spec = small_spec(synth, "C1x4", 16, 12)
    ph, gr, _ = synth.render(spec, "sphere")
    res = oracle.pipeline(spec, ph, gr, want=("z", "x", "y"))
    z = res["z"].copy()
    z[3, 5] = 0.0                                                 # an invalid pixel is dropped
    z[4, 5] = spec["fov_max"]                                     # the bounds themselves are kept
    pts = oracle.point_cloud(spec, z)
    keep = ~((z < spec["fov_min"]) | (z > spec["fov_max"]))
    assert pts.shape == (int(keep.sum()), 3)
    vv, uu = np.nonzero(keep.T)                                   # column-major walk
    assert np.array_equal(pts[:, 2], z.T[keep.T])
    cam = spec["calib"]["cam"]
    assert np.array_equal(pts[:, 0], z.T[keep.T] * (vv - cam[2]) / cam[0])
    assert np.array_equal(pts[:, 1], z.T[keep.T] * (uu - cam[5]) / cam[4])


# ---------------------------------------------------------------- dynamic frames (CalculateOther)
def test_strip_regression_against_numpy(oracle):
    """StripRegression restated independently: 21-row column sums by cumulative sums, extrema by argmax/argmin with the
    reference's tie rule (the centre wins, then the leftmost)."""
    rng = np.random.default_rng(0)
    cam = rng.integers(0, 256, (50, 70), dtype=np.uint8)
    cam[:, 30:40] = (np.arange(10) * 25)[None, :]                 # ties and ramps
    sw, sb = oracle.strip_regression(cam, 21)
    H, W, hw = 50, 70, 10
    vs = np.zeros((H, W))
    cs = np.cumsum(np.vstack([np.zeros((1, W)), cam.astype(np.float64)]), axis=0)
    for h in range(hw, H - hw):
        vs[h, hw:W - hw] = (cs[h + hw + 1] - cs[h - hw])[hw:W - hw]
    for h in range(H):
        for w in range(W):
            if not (hw <= h < H - hw and hw <= w < W - hw):
                assert sw[h, w] == 0 and sb[h, w] == 0
                continue
            seg, c = vs[h, w - hw:w + hw], vs[h, w]
            assert sw[h, w] == (0 if seg.max() <= c else int(np.argmax(seg)) - hw)
            assert sb[h, w] == (0 if seg.min() >= c else int(np.argmin(seg)) - hw)


def test_delta_p_selection_and_blur(oracle):
    from scipy.ndimage import uniform_filter
    rng = np.random.default_rng(1)
    W0, B0, W1, B1 = (rng.integers(-10, 10, (17, 23)).astype(np.float32) for _ in range(4))
    raw = np.where(np.abs(B0 - B1) < np.abs(W0 - W1), B0 - B1, W0 - W1)
    want = uniform_filter(raw.astype(np.float64), size=3, mode="mirror")      # BORDER_REFLECT_101
    got = oracle.delta_p(W0, B0, W1, B1)
    assert got.dtype == np.float32 and np.allclose(got, want, rtol=0, atol=2e-6)
    one = oracle.delta_p(*(a[:1, :1] for a in (W0, B0, W1, B1)))               # 1x1: every tap is the pixel itself
    assert one[0, 0] == np.float32(float(raw[0, 0]) * 9 * (1.0 / 9))


# ---------------------------------------------------------------- regression vectors
@pytest.mark.parametrize("name", GOLDEN_SCENES)
def test_scene_golden(oracle, synth, golden_dir, name):
    d = np.load(os.path.join(golden_dir, "scene_%s.npz" % name))
    spec = small_spec(synth, name)
    want = tuple(k[4:] for k in d.files if k.startswith("out_"))
    res = oracle.pipeline(spec, d["phase"] if "phase" in d.files else None,
                          d["gray_planes"] if "gray_planes" in d.files else None, want=want)
    for w in want:
        assert np.array_equal(res[w], d["out_" + w]), w


# ---------------------------------------------------------------- structure of the oracle itself
@pytest.mark.parametrize("name", ["C1x4", "C3", "C5"])
def test_threads_and_order_do_not_change_results(oracle, synth, name):
    spec = small_spec(synth, name, 75, 41)
    ph, gr = synth.random_planes(spec, seed=7)
    want = ("z", "x", "y", "U") + (("k",) if spec["n_freq"] > 1 else ()) + (("mask", "gray") if spec["gray_bits"] else ())
    a = oracle.pipeline(spec, ph, gr, want=want, threads=1, faithful_order=1)
    b = oracle.pipeline(spec, ph, gr, want=want, threads=3, faithful_order=0)
    for w in want:
        assert np.array_equal(a[w], b[w], equal_nan=True), w


def test_row_tile_equals_slice(oracle, synth, shard):
    spec = small_spec(synth, "C3", 64, 50)
    ph, gr = synth.random_planes(spec, seed=11)
    full = oracle.pipeline(spec, ph, gr, want=("z", "y", "mask"))
    for rank in range(3):
        tile, lo, hi = shard.row_tile_spec(spec, 3, rank)
        part = oracle.pipeline(tile, ph[:, lo:hi], gr[:, lo:hi], want=("z", "y", "mask"))
        for w in ("z", "y", "mask"):
            assert np.array_equal(part[w], full[w][lo:hi], equal_nan=True), (w, rank)


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (3, 2), (5, 511)])
def test_ragged_sizes(oracle, synth, shape):
    h, w = shape
    spec = small_spec(synth, "C3", w, h)
    ph, gr = synth.random_planes(spec, seed=h * 1000 + w)
    res = oracle.pipeline(spec, ph, gr, want=("z", "U", "mask"))
    assert res["z"].shape == (h, w)
    assert np.all((res["z"] == 0) | ((res["z"] >= spec["fov_min"]) & (res["z"] <= spec["fov_max"])))


def test_bad_config_rejected(oracle, synth):
    spec = small_spec(synth, "C2")
    ph, _ = synth.random_planes(spec, seed=1)
    with pytest.raises(ValueError):
        oracle.pipeline(dict(spec, n_steps=0), ph, None)
    with pytest.raises(ValueError):
        oracle.pipeline(dict(spec, periods=[0, 1, 2]), ph, None)


def test_reference_static_parameters(api, synth, golden_dir):
    """The reference's compiled-in configuration (R/StaticParameters.cpp) -- the one translation unit of the reference that compiles
    here without OpenCV.  tests/golden/static_parameters.json holds the constants read out of the COMPILED unit
    (oracle/_ref/libdynaframe_static.so, `make -C oracle ref`, tests/golden/make_static_parameters.py).  Pinned against them:
    the live shared object wherever it exists (this container; a GPU box that received it), the REF configuration of the tests and
    the bench, the defaults of the product's C++ mirror classes (slx_reference_defaults), and the two integer divisions the host
    loop derives from them (R/CCalculation.cpp:550, :562-563).  This pins CONFIGURATION, not arithmetic: parity stays unpinned."""
    import json
    import os
    import sys
    gold = json.load(open(os.path.join(golden_dir, "static_parameters.json")))["constants"]
    sys.path.insert(0, os.path.join(os.path.dirname(golden_dir), "..", "oracle"))
    import ref_static
    if ref_static.available():
        assert ref_static.constants() == gold                       # the fixture IS what the compiled reference unit holds
    ref = synth.make_spec("REF")
    assert (ref["width"], ref["height"], ref["proj_width"]) == (gold["CAMERA_RESLINE"], gold["CAMERA_RESROW"], gold["PROJECTOR_RESLINE"])
    assert (ref["gray_bits"], ref["n_steps"], ref["n_freq"]) == (gold["GRAY_V_NUMDIGIT"], gold["PHASE_NUMDIGIT"], 1)
    assert ref["periods"] == [gold["PROJECTOR_RESLINE"] // (1 << (gold["GRAY_V_NUMDIGIT"] - 1))] == [40]      # `1 << G - 1`, R/CCalculation.cpp:550
    assert ref["gray_stripe"] == gold["PROJECTOR_RESLINE"] // (1 << gold["GRAY_V_NUMDIGIT"]) == 20             # :562-563
    # the FOV window of the synthetic REF scene is the reference's, times the 10 x that puts its depths in millimetres
    assert (ref["fov_min"], ref["fov_max"]) == (10.0 * gold["FOV_MIN_DISTANCE"], 10.0 * gold["FOV_MAX_DISTANCE"])
    mine = api.reference_defaults()
    assert mine == {k: gold[k] for k in mine}, (mine, gold)
