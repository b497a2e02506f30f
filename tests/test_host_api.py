"""CPU tests of the boundary: libslx.so loads, exports every symbol include/slx.h declares,
validates configurations like the reference's setters do, and fails loudly (no CPU fallback)
when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "slx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slx_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(api):
    L = api.lib()
    declared = header_symbols()
    assert sorted(api.SYMBOLS) == declared
    for name in declared:
        assert hasattr(L, name), name
    assert L.slx_version() == 1


def test_cpp_mirror_classes_are_exported(api):
    out = subprocess.check_output(["nm", "-DC", api.LIB_PATH]).decode()
    for sym in ("slx::CDecodePhase::SetNumMat(int, int)", "slx::CDecodePhase::Decode()",
                "slx::CDecodeGray::SetNumDigit(int, bool)", "slx::CDecodeGray::Decode()",
                "slx::CCalculation::CalculateFirst()", "slx::ReadGrayCodeFile"):
        assert sym in out, sym


def test_no_oracle_in_product(api):
    """The product never links or imports anything under oracle/."""
    out = subprocess.check_output(["ldd", api.LIB_PATH]).decode()
    assert "oracle" not in out
    pkg_dir = os.path.dirname(api.LIB_PATH)
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"import\s+oracle|from\s+oracle|slx_oracle|slxo_|oracle/|oracle\.py", text), f


@pytest.mark.parametrize("name", ["C1", "C1x4", "REF", "C2", "C3", "C4", "C5"])
def test_named_configs_validate(api, synth, name):
    spec = synth.make_spec(name)
    rc, msg = api.validate_config(api.make_config(spec, aux=("U",)))
    assert rc == api.OK, msg
    n_phase, n_gray = synth.n_planes(spec)
    assert synth.algorithmic_bytes_per_pixel(spec) == n_phase + n_gray + 8


def test_baseline_byte_model(synth):
    # C2/C4 20 B/px, C3 32 B/px, C5 40 B/px, C1 12 / 24 B/px
    got = {n: synth.algorithmic_bytes_per_pixel(synth.make_spec(n)) for n in ("C1", "C1x4", "C2", "C3", "C4", "C5")}
    assert got == {"C1": 12, "C1x4": 24, "C2": 20, "C3": 32, "C4": 20, "C5": 40}
    s = synth.make_spec("C4")
    assert s["width"] * s["height"] * 20 == 46080000


BAD = [
    (dict(width=0), "width/height"),
    (dict(height=-3), "width/height"),
    (dict(mode=9), "unknown mode"),
    (dict(n_steps=0), "n_steps"),            # R/CDecodePhase.cpp:122
    (dict(n_steps=17), "n_steps"),
    (dict(n_freq=0), "n_freq"),
    (dict(n_freq=5), "n_freq"),
    (dict(periods=[1920, 0, 30]), "period[1]"),
    (dict(fov_min=5.0, fov_max=1.0), "fov_min"),
]


@pytest.mark.parametrize("patch,needle", BAD)
def test_bad_configs_are_rejected(api, synth, patch, needle):
    spec = dict(synth.make_spec("C4"), **patch)
    rc, msg = api.validate_config(api.make_config(spec))
    assert rc == api.ERR_INVALID_ARG
    assert needle in msg, msg


def test_bad_gray_configs_are_rejected(api, synth):
    spec = synth.make_spec("C3")
    for patch, needle in ((dict(gray_bits=0), "gray_bits"), (dict(gray_bits=17), "gray_bits"),   # R/CDecodeGray.cpp:39
                          (dict(gray_stripe=0), "gray_stripe"), (dict(gray_lut=None), "gray_lut")):
        rc, msg = api.validate_config(api.make_config(dict(spec, **patch)))
        assert rc == api.ERR_INVALID_ARG and needle in msg, (patch, msg)
    # an output the mode does not produce
    rc, msg = api.validate_config(api.make_config(synth.make_spec("C4"), aux=("gray",)))
    assert rc == api.ERR_INVALID_ARG and "aux_outputs" in msg
    rc, msg = api.validate_config(api.make_config(synth.make_spec("C1x4"), aux=("k",)))
    assert rc == api.ERR_INVALID_ARG
    rc, msg = api.validate_config(api.make_config(dict(synth.make_spec("C1x4"), n_freq=2, periods=[40, 20])))
    assert rc == api.ERR_INVALID_ARG and "exactly one frequency" in msg


def test_null_arguments(api):
    L = api.lib()
    assert L.slx_validate_config(None, None, 0) == api.ERR_INVALID_ARG
    assert L.slx_create(None, None) == api.ERR_INVALID_ARG
    assert L.slx_decode(None, None) == api.ERR_INVALID_ARG
    assert L.slx_set_frame(None, 0, 0, None, 0, 0) == api.ERR_INVALID_ARG
    L.slx_destroy(None)                                  # no-op, must not crash
    assert isinstance(L.slx_last_error(None), bytes)
    import ctypes as C
    p, n, m = C.c_void_p(), C.c_size_t(7), C.c_size_t(7)
    assert L.slx_get_point_cloud_text(None, C.byref(p), C.byref(n), C.byref(m)) == api.ERR_INVALID_ARG
    assert L.slx_format_points_text(None, None, 0, C.byref(p), C.byref(n)) == api.ERR_INVALID_ARG
    assert L.slx_get_point_cloud_view(None, C.byref(p), C.byref(n)) == api.ERR_INVALID_ARG


def test_create_fails_loudly_without_gpu(api, synth):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.SlxError) as e:
        api.Context(synth.make_spec("C4"))
    assert e.value.code == api.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_split_range(shard):
    for n in (0, 1, 7, 32, 256, 1200):
        for world in (1, 2, 3, 8):
            spans = [shard.split_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.split_range(4, 2, 2)
    assert shard.split_range(1200, 8, 3) == (450, 600)   # 150-row tiles of SURVEY.md section 8(e)


def test_row_tile_spec(shard, synth):
    spec = synth.make_spec("C4")
    tile, lo, hi = shard.row_tile_spec(spec, 8, 7)
    assert (lo, hi) == (1050, 1200) and tile["height"] == 150 and tile["row_offset"] == 1050
    assert spec["height"] == 1200 and spec["row_offset"] == 0


def test_synthetic_source_is_deterministic(synth):
    spec = dict(synth.make_spec("C3"), width=64, height=48)
    a = synth.render(spec, "sphere", seed=5, noise_sigma=2.0)
    b = synth.render(spec, "sphere", seed=5, noise_sigma=2.0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[0].shape == (12, 48, 64) and a[1].shape == (12, 48, 64)
    assert a[0].dtype == np.uint8


# ---------------------------------------------------------------- file formats of a data directory
def test_bmp_reader(api, tmp_path):
    from dynaframe_files import write_bmp
    rng = np.random.default_rng(3)
    for w, h in ((7, 5), (64, 48), (1, 1), (130, 3)):            # widths that need row padding included
        img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        for bits, top_down in ((8, False), (8, True), (24, False), (24, True)):
            path = str(tmp_path / ("a_%d_%d_%d_%d.bmp" % (w, h, bits, top_down)))
            write_bmp(path, img, bits=bits, top_down=top_down)
            assert np.array_equal(api.read_bmp_gray(path), img), (w, h, bits, top_down)
    # a palette that is not the identity: pixel values are indices (cv::imread goes through the palette)
    pal = np.arange(256, dtype=np.uint8)[::-1].copy()
    img = rng.integers(0, 256, size=(9, 11), dtype=np.uint8)
    write_bmp(str(tmp_path / "p.bmp"), img, palette=pal)
    assert np.array_equal(api.read_bmp_gray(str(tmp_path / "p.bmp")), pal[img])
    with pytest.raises(api.SlxError):
        api.read_bmp_gray(str(tmp_path / "missing.bmp"))
    (tmp_path / "junk.bmp").write_bytes(b"BM" + b"\0" * 10)
    with pytest.raises(api.SlxError):
        api.read_bmp_gray(str(tmp_path / "junk.bmp"))


def test_pgm_reader(api, tmp_path):
    rng = np.random.default_rng(5)
    for w, h in ((7, 5), (64, 48), (1, 1)):
        img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        for header in ("P5\n%d %d\n255\n" % (w, h), "P5 # made by a camera SDK\n# second comment\n%d\t%d\r\n255 " % (w, h)):
            path = str(tmp_path / "a.pgm")
            with open(path, "wb") as f:
                f.write(header.encode() + img.tobytes())
            assert np.array_equal(api.read_pgm_gray(path), img), (w, h, header)
    for bad in (b"P2\n2 2\n255\n1 2 3 4", b"P5\n2 2\n65535\n" + b"\0" * 8, b"P5\n4 4\n255\n" + b"\0" * 15, b"P5\n-2 2\n255\n"):
        (tmp_path / "bad.pgm").write_bytes(bad)
        with pytest.raises(api.SlxError):
            api.read_pgm_gray(str(tmp_path / "bad.pgm"))
    with pytest.raises(api.SlxError):
        api.read_pgm_gray(str(tmp_path / "missing.pgm"))


def test_bmp_colour_to_grey_matches_opencv_formula(api, tmp_path):
    import struct
    rng = np.random.default_rng(4)
    h, w = 6, 10
    bgr = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    row = (3 * w + 3) // 4 * 4
    body = np.zeros((h, row), dtype=np.uint8)
    body[:, : 3 * w] = bgr.reshape(h, 3 * w)
    path = str(tmp_path / "c.bmp")
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + body.size, 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, -h, 1, 24, 0, body.size, 2835, 2835, 0, 0))
        f.write(body.tobytes())
    b, g, r = (bgr[..., i].astype(np.uint32) for i in range(3))
    want = ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)     # OpenCV 2.4 BGR2GRAY, 8-bit
    assert np.array_equal(api.read_bmp_gray(path), want)


def test_calibration_yaml_reader(api, synth, tmp_path, golden_dir):
    import json
    from dynaframe_files import write_calibration_yaml
    yml = json.load(open(os.path.join(golden_dir, "result_yml.json")))
    path = str(tmp_path / "parameters.yml")
    write_calibration_yaml(path, yml["CamMat"], yml["ProMat"], yml["R"], yml["T"])
    text = open(path).read()
    assert text.startswith("%YAML:1.0") and "!!opencv-matrix" in text and "1.2138714552009253e+003" in text
    got = api.read_calibration_yaml(path)
    assert got["cam"] == yml["CamMat"] and got["pro"] == yml["ProMat"] and got["rot"] == yml["R"] and got["trans"] == yml["T"]
    with pytest.raises(api.SlxError):
        api.read_calibration_yaml(str(tmp_path / "none.yml"))
    (tmp_path / "short.yml").write_text(text.replace("ProMat", "Other"))
    with pytest.raises(api.SlxError):
        api.read_calibration_yaml(str(tmp_path / "short.yml"))


def test_point_cloud_text_writer(api, tmp_path):
    """CCalculation::Result's file (R/CCalculation.cpp:351-353: `file << x << ' ' << y << ' ' << z << endl`): operator<< prints a double
    as printf("%g") does, and so does Python's "%g" for finite values.  (The C++ test in tests/cpp/host_sanitize.cpp compares with
    operator<< itself, NaNs and all, under ASan.)"""
    rng = np.random.default_rng(7)
    xyz = (rng.random((70001, 3)) - 0.3) * 1500.0
    xyz[::5] = rng.standard_normal((xyz[::5].shape[0], 3)) * 10.0 ** rng.integers(-12, 12, size=(xyz[::5].shape[0], 1))
    xyz[:6] = [[0.0, -0.0, 1.0], [999999.5, 999999.4, 0.0001], [0.00009999995, 123456.5, 1234565.0], [1e-300, -1e300, 5e-324],
               [np.inf, -np.inf, 2.5], [100000.5, 99999.95, 123.4565]]
    path = str(tmp_path / "cloud.txt")
    api.write_point_cloud_text(path, xyz)
    want = "".join("%g %g %g\n" % tuple(p) for p in xyz)
    assert open(path).read() == want
    api.write_point_cloud_text(path, np.empty((0, 3)))
    assert open(path).read() == ""
    with pytest.raises(ValueError):
        api.write_point_cloud_text(path, np.zeros((4, 2)))
    with pytest.raises(api.SlxError):
        api.write_point_cloud_text(str(tmp_path / "no" / "such" / "dir" / "cloud.txt"), xyz[:1])
    # SLX_TEXT_MSVC2013: the bytes of the reference as built (MSVC 2013 runtime, `fstream` in text mode): the same digits, at least
    # three exponent digits, that runtime's spellings of the non-finite values, CR LF
    import re
    api.write_point_cloud_text(path, xyz, dialect=api.TEXT_MSVC2013)
    got = open(path, "rb").read().decode()
    finite = "".join("%g %g %g\r\n" % tuple(p) for p in xyz).replace("-inf", "-1.#INF").replace("inf", "1.#INF")
    assert got == re.sub(r"e([+-])(\d\d)(?!\d)", r"e\g<1>0\2", finite)
    assert got.startswith("0 -0 1\r\n1e+006 999999 0.0001\r\n0.0001 123456 1.23456e+006\r\n1e-300 -1e+300 4.94066e-324\r\n1.#INF -1.#INF 2.5\r\n")
    api.write_point_cloud_text(path, np.array([[np.nan, -np.nan, 5e-5]]), dialect=api.TEXT_MSVC2013)
    assert open(path, "rb").read() in (b"1.#QNAN -1.#IND 5e-005\r\n", b"-1.#IND 1.#QNAN 5e-005\r\n")   # (numpy may flip the sign bit of its NaN constant)
    with pytest.raises(api.SlxError):
        api.write_point_cloud_text(path, xyz[:2], dialect=9)
