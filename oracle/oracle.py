"""ctypes binding of the CPU oracle (oracle/libslx_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.  PARITY STATUS:
"parity unpinned" (see slx_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libslx_oracle.so")

MODE_PHASE_ONLY, MODE_GRAY_ONLY, MODE_GRAY_PHASE, MODE_MULTIFREQ, MODE_MULTIFREQ_GRAYMASK = range(5)
MAX_FREQ = 4
MAX_STEPS = 16


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("row_offset", C.c_int), ("col_offset", C.c_int),
        ("mode", C.c_int), ("n_freq", C.c_int), ("n_steps", C.c_int),
        ("period", C.c_int * MAX_FREQ),
        ("gray_bits", C.c_int), ("gray_stripe", C.c_int),
        ("gray_lut", C.POINTER(C.c_int16)),
        ("fov_min", C.c_double), ("fov_max", C.c_double),
        ("cam", C.c_double * 9), ("pro", C.c_double * 9),
        ("rot", C.c_double * 9), ("trans", C.c_double * 3),
        ("faithful_order", C.c_int),
    ]


class Outputs(C.Structure):
    _fields_ = [
        ("z", C.POINTER(C.c_double)), ("x", C.POINTER(C.c_double)),
        ("y", C.POINTER(C.c_double)), ("U", C.POINTER(C.c_double)),
        ("pix", C.POINTER(C.c_double)), ("gray", C.POINTER(C.c_double)),
        ("k", C.POINTER(C.c_int32)), ("mask", C.POINTER(C.c_uint8)),
    ]


def build(force=False):
    """Compile the oracle with its Makefile (gcc)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "slx_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.slxo_fast_atan2_deg.restype = C.c_float
        L.slxo_fast_atan2_deg.argtypes = [C.c_float, C.c_float]
        L.slxo_pipeline.restype = C.c_int
        L.slxo_triangulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double] + [C.c_void_p] * 5
        L.slxo_calib_tables.argtypes = [C.c_void_p] * 5
        L.slxo_pipeline_mt.restype = C.c_int
        _lib = L
    return _lib


def fast_atan2_deg(y, x):
    return float(lib().slxo_fast_atan2_deg(float(y), float(x)))


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def make_config(spec, faithful_order=0, row_offset=None, height=None):
    """spec: a dict as produced by the package's synth.make_spec()."""
    c = Config()
    c.width = spec["width"]
    c.height = spec["height"] if height is None else height
    c.row_offset = spec.get("row_offset", 0) if row_offset is None else row_offset
    c.col_offset = 0
    c.mode = spec["mode"]
    c.n_freq = spec.get("n_freq", 1)
    c.n_steps = spec.get("n_steps", 4)
    for i, t in enumerate(spec.get("periods", [])):
        c.period[i] = int(t)
    c.gray_bits = spec.get("gray_bits", 0)
    c.gray_stripe = spec.get("gray_stripe", 0)
    lut = spec.get("gray_lut")
    keep = None
    if lut is not None:
        keep = np.ascontiguousarray(lut, dtype=np.int16)
        c.gray_lut = _ptr(keep, C.c_int16)
    c.fov_min = spec.get("fov_min", 0.0)
    c.fov_max = spec.get("fov_max", 0.0)
    cal = spec.get("calib")
    if cal is not None:
        for name, n in (("cam", 9), ("pro", 9), ("rot", 9), ("trans", 3)):
            v = np.asarray(cal[name], dtype=np.float64).reshape(-1)
            assert v.size == n
            for i in range(n):
                getattr(c, name)[i] = float(v[i])
    c.faithful_order = faithful_order
    c._keep = keep
    return c


def pipeline(spec, phase_planes, gray_planes=None, want=("z",), threads=1, faithful_order=0):
    """Run the whole oracle path on one frame-set.

    phase_planes: uint8 array [F*N, H, W] (or None); gray_planes: uint8 [2G, H, W] (or None).
    Returns a dict of numpy arrays for the names in `want`
    (z, x, y, U: f64 [H,W]; pix: f64 [F,H,W]; gray: f64 [H,W]; k: i32 [F-1,H,W]; mask: u8 [H,W]).
    """
    L = lib()
    H, W = spec["height"], spec["width"]
    F = spec.get("n_freq", 1)
    cfg = make_config(spec, faithful_order=faithful_order)
    stride = W

    def plane_ptrs(arr):
        if arr is None:
            return None, None
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        assert arr.shape[1:] == (H, W), (arr.shape, H, W)
        n = arr.shape[0]
        ptrs = (C.POINTER(C.c_uint8) * n)()
        for i in range(n):
            ptrs[i] = arr[i].ctypes.data_as(C.POINTER(C.c_uint8))
        return arr, ptrs

    pa, pp = plane_ptrs(phase_planes)
    ga, gp = plane_ptrs(gray_planes)
    out = Outputs()
    res = {}
    shapes = {
        "z": ((H, W), np.float64, C.c_double), "x": ((H, W), np.float64, C.c_double),
        "y": ((H, W), np.float64, C.c_double), "U": ((H, W), np.float64, C.c_double),
        "pix": ((F, H, W), np.float64, C.c_double), "gray": ((H, W), np.float64, C.c_double),
        "k": ((max(F - 1, 0), H, W), np.int32, C.c_int32), "mask": ((H, W), np.uint8, C.c_uint8),
    }
    for name in want:
        shp, dt, ct = shapes[name]
        a = np.zeros(shp, dtype=dt)
        res[name] = a
        if a.size:
            setattr(out, name, _ptr(a, ct))
    if threads == 1:
        rc = L.slxo_pipeline(C.byref(cfg), pp, gp, C.c_size_t(stride), C.byref(out))
    else:
        rc = L.slxo_pipeline_mt(C.byref(cfg), pp, gp, C.c_size_t(stride), C.byref(out), int(threads))
    if rc != 0:
        raise ValueError("oracle rejected the configuration: rc=%d" % rc)
    del pa, ga
    return res


def point_cloud(spec, z):
    """Packed (x, y, z) of the depths inside the FOV in CCalculation::Result's order: float64 [n, 3]."""
    L = lib()
    L.slxo_point_cloud.restype = C.c_size_t
    cfg = make_config(spec)
    z = np.ascontiguousarray(z, dtype=np.float64)
    xyz = np.zeros((z.size, 3))
    n = L.slxo_point_cloud(C.byref(cfg), _ptr(z, C.c_double), _ptr(xyz, C.c_double))
    return xyz[:n].copy()


def strip_regression(cam, win=21):
    """CCalculation::StripRegression: (stripW, stripB) float32 [H, W]."""
    cam = np.ascontiguousarray(cam, dtype=np.uint8)
    H, W = cam.shape
    sw, sb = np.zeros((H, W), dtype=np.float32), np.zeros((H, W), dtype=np.float32)
    lib().slxo_strip_regression(_ptr(cam, C.c_uint8), C.c_size_t(W), W, H, int(win), _ptr(sw, C.c_float), _ptr(sb, C.c_float))
    return sw, sb


def delta_p(W0, B0, W1, B1):
    """FillOtherDeltaProU up to and including the 3x3 blur: float32 [H, W]."""
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (W0, B0, W1, B1)]
    H, W = arrs[0].shape
    out = np.zeros((H, W), dtype=np.float32)
    lib().slxo_delta_p(*[_ptr(a, C.c_float) for a in arrs], W, H, _ptr(out, C.c_float))
    return out


def triangulate(spec, U, want=("z",)):
    """FillCoordinate on a given projector-column map."""
    L = lib()
    cfg = make_config(spec)
    H, W = spec["height"], spec["width"]
    U = np.ascontiguousarray(U, dtype=np.float64)
    cC, cD = np.zeros((H, W)), np.zeros((H, W))
    cA, cB = C.c_double(), C.c_double()
    L.slxo_calib_tables(C.byref(cfg), C.byref(cA), C.byref(cB), _ptr(cC, C.c_double), _ptr(cD, C.c_double))
    z, x, y = np.zeros((H, W)), np.zeros((H, W)), np.zeros((H, W))
    L.slxo_triangulate(C.byref(cfg), _ptr(U, C.c_double), None, cA, cB, _ptr(cC, C.c_double), _ptr(cD, C.c_double),
                       _ptr(z, C.c_double), _ptr(x, C.c_double), _ptr(y, C.c_double))
    return {"z": z, "x": x, "y": y}


def projection_matrix(pro, rot, trans):
    L = lib()
    P = np.zeros(12)
    L.slxo_projection_matrix(_ptr(np.ascontiguousarray(pro, dtype=np.float64).reshape(-1), C.c_double),
                             _ptr(np.ascontiguousarray(rot, dtype=np.float64).reshape(-1), C.c_double),
                             _ptr(np.ascontiguousarray(trans, dtype=np.float64).reshape(-1), C.c_double),
                             _ptr(P, C.c_double))
    return P.reshape(3, 4)


def gray_lut_from_rows(rows):
    rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 2)
    lut = np.zeros(rows.shape[0], dtype=np.int16)
    rc = lib().slxo_gray_lut_from_rows(_ptr(rows, C.c_int), rows.shape[0], _ptr(lut, C.c_int16))
    if rc != 0:
        raise ValueError("bad gray-code rows")
    return lut


def nstep_weights(n):
    wy = np.zeros(MAX_STEPS, dtype=np.float32)
    wx = np.zeros(MAX_STEPS, dtype=np.float32)
    sc = C.c_float()
    lib().slxo_nstep_weights(int(n), _ptr(wy, C.c_float), _ptr(wx, C.c_float), C.byref(sc))
    return wy[:n].copy(), wx[:n].copy(), float(sc.value)


def wrapped_phase_generic(planes, period):
    """x1 generic path at any N (test hook)."""
    planes = np.ascontiguousarray(planes, dtype=np.uint8)
    n, H, W = planes.shape
    ptrs = (C.POINTER(C.c_uint8) * n)()
    for i in range(n):
        ptrs[i] = planes[i].ctypes.data_as(C.POINTER(C.c_uint8))
    out = np.zeros((H, W))
    lib().slxo_wrapped_phase_generic(ptrs, n, C.c_size_t(W), W, H, int(period), _ptr(out, C.c_double))
    return out
