"""ctypes binding of libslx.so -- the C ABI declared in include/slx.h.

This is the Python host side of the boundary: it adds nothing to the data path
(PyTorch / numpy only carry buffers).  If libslx.so is missing the import of
`lib()` raises: there is no fallback implementation.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libslx.so")

MAX_FREQ, MAX_STEPS, MAX_GRAY_BITS = 4, 16, 16

OK = 0
ERR_INVALID_ARG, ERR_NOT_CONFIGURED, ERR_MISSING_FRAME, ERR_NO_DEVICE = -1, -2, -3, -4
ERR_HIP, ERR_OUT_OF_MEMORY, ERR_NOT_DECODED, ERR_UNAVAILABLE = -5, -6, -7, -8

MODE_PHASE_ONLY, MODE_GRAY_ONLY, MODE_GRAY_PHASE, MODE_MULTIFREQ, MODE_MULTIFREQ_GRAYMASK = range(5)
MEM_HOST, MEM_DEVICE = 0, 1
GROUP_GRAY, GROUP_PHASE = 0, 1
OUT_Z, OUT_X, OUT_Y, OUT_U, OUT_PIX, OUT_GRAY, OUT_K, OUT_MASK, OUT_DELTAZ, OUT_DELTAP, OUT_STRIPW, OUT_STRIPB = range(12)
OUT_NAMES = {"z": OUT_Z, "x": OUT_X, "y": OUT_Y, "U": OUT_U, "pix": OUT_PIX, "gray": OUT_GRAY, "k": OUT_K, "mask": OUT_MASK,
             "deltaZ": OUT_DELTAZ, "deltaP": OUT_DELTAP, "stripW": OUT_STRIPW, "stripB": OUT_STRIPB}

# every symbol include/slx.h declares
SYMBOLS = [
    "slx_validate_config", "slx_create", "slx_destroy", "slx_last_error", "slx_set_gray_lut",
    "slx_set_frame", "slx_decode", "slx_decode_batch", "slx_synchronize", "slx_get_stream", "slx_get_output",
    "slx_get_depth", "slx_get_point_cloud", "slx_point_cloud_of_depth", "slx_track_begin", "slx_track_next", "slx_track_image_buffer", "slx_track_next_batch", "slx_track_stage_frames", "slx_track_frames_buffer", "slx_output_device_ptr", "slx_get_calibration", "slx_enable_timing",
    "slx_last_decode_ms", "slx_debug_stamps", "slx_set_variant", "slx_set_tuning", "slx_last_kernel", "slx_read_bmp_gray", "slx_read_pgm_gray", "slx_read_calibration_yaml", "slx_write_point_cloud_text", "slx_write_point_cloud_text_ex", "slx_set_text_dialect", "slx_get_point_cloud_view", "slx_get_point_cloud_text", "slx_format_points_text", "slx_version",
    "slx_decode_batch_ex", "slx_comm_unique_id", "slx_comm_create", "slx_comm_adopt", "slx_comm_destroy", "slx_comm_info", "slx_comm_last_error",
    "slx_comm_synchronize", "slx_gather_depth", "slx_decode_gather", "slx_gather_plan", "slx_gather_plan_ex", "slx_comm_set_gather_shape", "slx_scatter_rows", "slx_reference_defaults",
    "slx_pipe_create", "slx_pipe_destroy", "slx_pipe_layout", "slx_pipe_acquire", "slx_pipe_submit", "slx_pipe_collect", "slx_pipe_last_error",
]


class SlxConfig(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("row_offset", C.c_int), ("mode", C.c_int),
        ("n_freq", C.c_int), ("n_steps", C.c_int), ("period", C.c_int * MAX_FREQ),
        ("gray_bits", C.c_int), ("gray_stripe", C.c_int),
        ("gray_lut", C.POINTER(C.c_int16)),
        ("fov_min", C.c_double), ("fov_max", C.c_double),
        ("cam", C.c_double * 9), ("pro", C.c_double * 9), ("rot", C.c_double * 9), ("trans", C.c_double * 3),
        ("device", C.c_int), ("aux_outputs", C.c_uint),
    ]


class SlxBatchOut(C.Structure):
    _fields_ = [("z", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p), ("U", C.c_void_p), ("k", C.c_void_p), ("mask", C.c_void_p),
                ("plane_stride", C.c_size_t)]


class SlxShard(C.Structure):
    _fields_ = [("set0", C.c_int), ("n_sets", C.c_int), ("row0", C.c_int), ("rows", C.c_int)]


class SlxMsg(C.Structure):
    _fields_ = [("peer", C.c_int), ("send", C.c_int), ("offset", C.c_ulonglong), ("count", C.c_ulonglong)]


class SlxScatter(C.Structure):
    _fields_ = [("src", C.c_ulonglong), ("dst", C.c_ulonglong), ("run", C.c_ulonglong), ("n_runs", C.c_ulonglong), ("src_stride", C.c_ulonglong),
                ("dst_stride", C.c_ulonglong)]


COMM_ID_BYTES = 128
GATHER_IN_PLACE, GATHER_STAGED = 0, 1
GATHER_SHAPES = {"in_place": GATHER_IN_PLACE, "staged": GATHER_STAGED}


class SlxPipeConfig(C.Structure):
    _fields_ = [("slots", C.c_int), ("sets_per_slot", C.c_int), ("host_result", C.c_int)]


class SlxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("slx error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Loads libslx.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libslx.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no fallback implementation)")
        try:
            # When PyTorch is installed, its wheel's HIP runtime and RCCL (same SONAMEs as /opt/rocm's) must be the copies this
            # process binds, whichever of the two libraries is asked for first: buffers and streams cross between them.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, sz = C.c_void_p, C.c_size_t
        L.slx_validate_config.argtypes = [C.POINTER(SlxConfig), C.c_char_p, sz]
        L.slx_create.argtypes = [C.POINTER(SlxConfig), C.POINTER(vp)]
        L.slx_destroy.argtypes = [vp]
        L.slx_destroy.restype = None
        L.slx_last_error.argtypes = [vp]
        L.slx_last_error.restype = C.c_char_p
        L.slx_set_gray_lut.argtypes = [vp, C.POINTER(C.c_int16), sz]
        L.slx_set_frame.argtypes = [vp, C.c_int, C.c_int, vp, sz, C.c_int]
        L.slx_decode.argtypes = [vp, vp]
        L.slx_decode_batch.argtypes = [vp, C.c_int, vp, sz, vp, sz, sz, vp, vp]
        L.slx_decode_batch_ex.argtypes = [vp, C.c_int, vp, sz, vp, sz, sz, C.POINTER(SlxBatchOut), vp]
        L.slx_comm_unique_id.argtypes = [vp, sz]
        L.slx_comm_create.argtypes = [vp, vp, sz, C.c_int, C.c_int, C.POINTER(vp)]
        L.slx_comm_adopt.argtypes = [vp, vp, C.POINTER(vp)]
        L.slx_comm_destroy.argtypes = [vp]
        L.slx_comm_destroy.restype = None
        L.slx_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.slx_comm_last_error.argtypes = [vp]
        L.slx_comm_last_error.restype = C.c_char_p
        L.slx_comm_synchronize.argtypes = [vp]
        L.slx_gather_depth.argtypes = [vp, C.POINTER(SlxShard), C.c_int, C.c_int, vp, sz, vp, C.c_int, vp]
        L.slx_decode_gather.argtypes = [vp, vp, C.POINTER(SlxShard), C.c_int, C.c_int, vp, sz, vp, sz, sz, vp, vp, C.c_int, vp]
        L.slx_gather_plan.argtypes = [C.POINTER(SlxShard), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, sz, C.c_int, C.POINTER(SlxMsg), C.c_int,
                                      C.POINTER(C.c_int)]
        L.slx_gather_plan_ex.argtypes = [C.POINTER(SlxShard), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, sz, C.c_int, C.c_int, C.POINTER(SlxMsg), C.c_int,
                                         C.POINTER(C.c_int), C.POINTER(SlxScatter), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong)]
        L.slx_comm_set_gather_shape.argtypes = [vp, C.c_int]
        L.slx_scatter_rows.argtypes = [vp, C.POINTER(SlxScatter), C.c_int, vp, vp, vp]
        L.slx_reference_defaults.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
        L.slx_synchronize.argtypes = [vp]
        L.slx_get_stream.argtypes = [vp, C.POINTER(vp)]
        L.slx_get_output.argtypes = [vp, C.c_int, vp, sz, C.c_int]
        L.slx_get_depth.argtypes = [vp, vp, C.c_int]
        L.slx_get_point_cloud.argtypes = [vp, vp, sz, C.POINTER(sz), C.c_int]
        L.slx_point_cloud_of_depth.argtypes = [vp, vp, vp, sz, C.POINTER(sz), C.c_int]
        L.slx_get_point_cloud_view.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
        L.slx_get_point_cloud_text.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
        L.slx_format_points_text.argtypes = [vp, vp, sz, C.POINTER(vp), C.POINTER(sz)]
        L.slx_track_begin.argtypes = [vp, vp, sz, C.c_int, C.c_int]
        L.slx_track_next.argtypes = [vp, vp, sz, C.c_int]
        L.slx_track_next_batch.argtypes = [vp, vp, sz, sz, C.c_int, C.c_int, vp, C.c_int]
        L.slx_track_stage_frames.argtypes = [vp, vp, sz, sz, C.c_int, C.POINTER(vp)]
        L.slx_track_frames_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
        L.slx_track_image_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
        L.slx_output_device_ptr.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.slx_get_calibration.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.slx_enable_timing.argtypes = [vp, C.c_int]
        L.slx_last_decode_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.slx_set_variant.argtypes = [vp, C.c_int]
        L.slx_set_tuning.argtypes = [vp, C.c_int, C.c_int]
        L.slx_last_kernel.argtypes = [vp, C.c_char_p, sz]
        L.slx_debug_stamps.argtypes = [vp, vp, sz]
        L.slx_read_bmp_gray.argtypes = [C.c_char_p, vp, sz, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.slx_read_pgm_gray.argtypes = [C.c_char_p, vp, sz, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.slx_read_calibration_yaml.argtypes = [C.c_char_p] + [C.POINTER(C.c_double)] * 4
        L.slx_write_point_cloud_text.argtypes = [C.c_char_p, vp, sz]
        L.slx_write_point_cloud_text_ex.argtypes = [C.c_char_p, vp, sz, C.c_int]
        L.slx_set_text_dialect.argtypes = [vp, C.c_int]
        L.slx_pipe_create.argtypes = [vp, C.POINTER(SlxPipeConfig), C.POINTER(vp)]
        L.slx_pipe_destroy.argtypes = [vp]
        L.slx_pipe_destroy.restype = None
        L.slx_pipe_layout.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
        L.slx_pipe_acquire.argtypes = [vp, C.POINTER(vp)]
        L.slx_pipe_submit.argtypes = [vp, C.c_int]
        L.slx_pipe_collect.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int)]
        L.slx_pipe_last_error.argtypes = [vp]
        L.slx_pipe_last_error.restype = C.c_char_p
        for name in SYMBOLS:
            if name not in ("slx_destroy", "slx_last_error", "slx_pipe_destroy", "slx_pipe_last_error", "slx_comm_destroy", "slx_comm_last_error"):
                getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


def make_config(spec, device=-1, aux=()):
    """spec: dict as produced by synth.make_spec() (width, height, mode, periods, ...)."""
    c = SlxConfig()
    c.width, c.height = spec["width"], spec["height"]
    c.row_offset = spec.get("row_offset", 0)
    c.mode = spec["mode"]
    c.n_freq = spec.get("n_freq", 1)
    c.n_steps = spec.get("n_steps", 4)
    for i, t in enumerate(spec.get("periods", [])[:MAX_FREQ]):
        c.period[i] = int(t)
    c.gray_bits = spec.get("gray_bits", 0)
    c.gray_stripe = spec.get("gray_stripe", 0)
    keep = None
    if spec.get("gray_lut") is not None:
        keep = np.ascontiguousarray(spec["gray_lut"], dtype=np.int16)
        c.gray_lut = keep.ctypes.data_as(C.POINTER(C.c_int16))
    c.fov_min = spec.get("fov_min", 0.0)
    c.fov_max = spec.get("fov_max", 0.0)
    cal = spec.get("calib")
    if cal is not None:
        for name, n in (("cam", 9), ("pro", 9), ("rot", 9), ("trans", 3)):
            v = np.asarray(cal[name], dtype=np.float64).reshape(-1)
            for i in range(n):
                getattr(c, name)[i] = float(v[i])
    c.device = device
    bits = 0
    for a in aux:
        bits |= 1 << (OUT_NAMES[a] if isinstance(a, str) else int(a))
    c.aux_outputs = bits
    c._keep = keep
    return c


def validate_config(cfg):
    buf = C.create_string_buffer(512)
    rc = lib().slx_validate_config(C.byref(cfg), buf, 512)
    return rc, buf.value.decode()


_OUT_DTYPE = {OUT_K: np.int32, OUT_MASK: np.uint8, OUT_DELTAP: np.float32, OUT_STRIPW: np.float32, OUT_STRIPB: np.float32}


class Context:
    """One decoder context (≙ the reference's decoder objects + result planes)."""

    def __init__(self, spec, device=-1, aux=()):
        self.spec = spec
        self.cfg = make_config(spec, device=device, aux=aux)
        self._h = C.c_void_p()
        rc = lib().slx_create(C.byref(self.cfg), C.byref(self._h))
        if rc != OK:
            raise SlxError(rc, lib().slx_last_error(None).decode())
        self._borrowed = []

    def close(self):
        if self._h:
            lib().slx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise SlxError(rc, lib().slx_last_error(self._h).decode())

    def last_error(self):
        return lib().slx_last_error(self._h).decode()

    def set_frame(self, group, idx, data, stride=None):
        """data: numpy uint8 [H,W] (host, deep-copied) or a torch CUDA uint8 tensor (borrowed)."""
        if isinstance(data, np.ndarray):
            assert data.dtype == np.uint8 and data.ndim == 2
            if data.strides[1] != 1:
                data = np.ascontiguousarray(data)
            self._check(lib().slx_set_frame(self._h, group, idx, data.ctypes.data, data.strides[0], MEM_HOST))
        else:  # torch tensor on the device
            assert data.is_cuda and data.dim() == 2 and data.stride(1) == 1
            self._borrowed.append(data)
            self._check(lib().slx_set_frame(self._h, group, idx, data.data_ptr(),
                                            data.stride(0) if stride is None else stride, MEM_DEVICE))

    def set_frames(self, phase=None, gray=None):
        self._borrowed = []
        if phase is not None:
            for i in range(phase.shape[0]):
                self.set_frame(GROUP_PHASE, i, phase[i])
        if gray is not None:
            for i in range(gray.shape[0]):
                self.set_frame(GROUP_GRAY, i, gray[i])

    def set_gray_lut(self, lut):
        lut = np.ascontiguousarray(lut, dtype=np.int16)
        self._check(lib().slx_set_gray_lut(self._h, lut.ctypes.data_as(C.POINTER(C.c_int16)), lut.size))

    def decode(self, stream=None):
        self._check(lib().slx_decode(self._h, stream))

    def decode_batch(self, n_sets, phase=None, gray=None, z_out=None, stream=None, row_stride=None):
        """phase: CUDA uint8 [n_sets, F*N, H, W]; gray: CUDA uint8 [n_sets, 2G, H, W]; z_out: CUDA f64 [n_sets,H,W]."""
        def base(t):
            if t is None:
                return None, 0
            assert t.is_cuda and t.stride(-1) == 1
            return t.data_ptr(), t.stride(0) * t.element_size()
        pb, ps = base(phase)
        gb, gs = base(gray)
        ref = phase if phase is not None else gray
        rs = ref.stride(-2) if row_stride is None else row_stride
        self._check(lib().slx_decode_batch(self._h, n_sets, pb, ps, gb, gs, rs, z_out.data_ptr(), stream))

    def decode_batch_ex(self, n_sets, phase=None, gray=None, z=None, x=None, y=None, U=None, k=None, mask=None, plane_stride=0, stream=None,
                        row_stride=None):
        """slx_decode_batch_ex: like decode_batch, with the optional outputs (CUDA tensors, or raw device addresses) and a
        plane stride in elements (0 = dense)."""
        def base(t):
            if t is None:
                return None, 0
            assert t.is_cuda and t.stride(-1) == 1
            return t.data_ptr(), t.stride(0) * t.element_size()

        def addr(t):
            return None if t is None else (t if isinstance(t, int) else t.data_ptr())
        pb, ps = base(phase)
        gb, gs = base(gray)
        ref = phase if phase is not None else gray
        rs = ref.stride(-2) if row_stride is None else row_stride
        out = SlxBatchOut(addr(z), addr(x), addr(y), addr(U), addr(k), addr(mask), int(plane_stride))
        self._check(lib().slx_decode_batch_ex(self._h, n_sets, pb, ps, gb, gs, rs, C.byref(out), stream))

    def synchronize(self):
        self._check(lib().slx_synchronize(self._h))

    def stream_handle(self):
        """The context's own hipStream_t as an integer (wrap it with torch.cuda.ExternalStream to record events on it)."""
        p = C.c_void_p()
        self._check(lib().slx_get_stream(self._h, C.byref(p)))
        return p.value

    def output_shape(self, which):
        H, W, F = self.spec["height"], self.spec["width"], self.spec.get("n_freq", 1)
        if which == OUT_PIX:
            return (F, H, W)
        if which == OUT_K:
            return (F - 1, H, W)
        return (H, W)

    def get_output(self, which):
        if isinstance(which, str):
            which = OUT_NAMES[which]
        a = np.empty(self.output_shape(which), dtype=_OUT_DTYPE.get(which, np.float64))
        self._check(lib().slx_get_output(self._h, which, a.ctypes.data, a.nbytes, MEM_HOST))
        return a

    def get_depth(self):
        a = np.empty(self.output_shape(OUT_Z), dtype=np.float64)
        self._check(lib().slx_get_depth(self._h, a.ctypes.data, MEM_HOST))
        return a

    def get_point_cloud(self):
        """Packed (x, y, z) of the depths inside the FOV, column-major order like CCalculation::Result: float64 [n, 3]."""
        n = C.c_size_t(0)
        rc = lib().slx_get_point_cloud(self._h, None, 0, C.byref(n), MEM_HOST)
        if n.value == 0:
            self._check(rc)
            return np.empty((0, 3), dtype=np.float64)
        a = np.empty((n.value, 3), dtype=np.float64)
        self._check(lib().slx_get_point_cloud(self._h, a.ctypes.data, n.value, C.byref(n), MEM_HOST))
        return a

    def get_point_cloud_text(self):
        """The cloud as the text CCalculation::Result writes, formatted on the device (slx_get_point_cloud_text): (bytes, number of points).
        SlxError ERR_UNAVAILABLE when a coordinate lies outside the device formatter's range."""
        p, nb, npts = C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
        self._check(lib().slx_get_point_cloud_text(self._h, C.byref(p), C.byref(nb), C.byref(npts)))
        return (C.string_at(p.value, nb.value) if nb.value else b""), npts.value

    def set_text_dialect(self, dialect):
        """TEXT_LIBSTDCXX or TEXT_MSVC2013 for get_point_cloud_text / format_points_text (slx_set_text_dialect)."""
        self._check(lib().slx_set_text_dialect(self._h, int(dialect)))

    def format_points_text(self, xyz):
        """Text of packed (x, y, z) triples in device memory (a CUDA f64 tensor [n, 3]), formatted on the device: bytes."""
        assert xyz.is_cuda and xyz.is_contiguous() and xyz.element_size() == 8 and xyz.numel() % 3 == 0
        p, nb = C.c_void_p(), C.c_size_t(0)
        self._check(lib().slx_format_points_text(self._h, xyz.data_ptr(), xyz.numel() // 3, C.byref(p), C.byref(nb)))
        return C.string_at(p.value, nb.value) if nb.value else b""

    def get_point_cloud_view(self):
        """The same cloud as a read-only numpy view [n, 3] of pinned memory the context owns (slx_get_point_cloud_view): valid until
        the next point-cloud call on this context."""
        p, n = C.c_void_p(), C.c_size_t(0)
        self._check(lib().slx_get_point_cloud_view(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.empty((0, 3), dtype=np.float64)
        a = np.ctypeslib.as_array((C.c_double * (n.value * 3)).from_address(p.value)).reshape(n.value, 3)
        a.flags.writeable = False
        return a

    def point_cloud_of_depth(self, depth, out=None):
        """Cloud of a depth plane in device memory (a CUDA f64 tensor [H, W], e.g. one frame-set of decode_batch's z).
        out: a CUDA f64 tensor [H*W, 3] to fill (returns the number of points), or None for a numpy array [n, 3]."""
        H, W = self.spec["height"], self.spec["width"]
        assert depth.is_cuda and depth.is_contiguous() and tuple(depth.shape) == (H, W) and depth.element_size() == 8
        n = C.c_size_t(0)
        if out is not None:
            assert out.is_cuda and out.is_contiguous() and out.element_size() == 8 and out.numel() % 3 == 0
            self._check(lib().slx_point_cloud_of_depth(self._h, depth.data_ptr(), out.data_ptr(), out.numel() // 3, C.byref(n), MEM_DEVICE))
            return n.value
        a = np.empty((H * W, 3), dtype=np.float64)
        self._check(lib().slx_point_cloud_of_depth(self._h, depth.data_ptr(), a.ctypes.data, H * W, C.byref(n), MEM_HOST))
        return a[:n.value].copy()

    def _image_args(self, image):
        if isinstance(image, np.ndarray):
            assert image.dtype == np.uint8 and image.ndim == 2 and image.strides[1] == 1
            return image.ctypes.data, image.strides[0], MEM_HOST
        assert image.is_cuda and image.dim() == 2 and image.stride(1) == 1
        self._borrowed.append(image)
        return image.data_ptr(), image.stride(0), MEM_DEVICE

    def track_image_buffer(self):
        """The pinned staging buffer of the next host-fed track call as a numpy uint8 [H, W] view: fill it and hand it back
        to track_begin / track_next, which then skips its own copy."""
        p, stride = C.c_void_p(), C.c_size_t()
        self._check(lib().slx_track_image_buffer(self._h, C.byref(p), C.byref(stride)))
        views = self.__dict__.setdefault("_track_views", {})         # two slots: build each numpy view once
        if p.value not in views:
            H, W = self.spec["height"], self.spec["width"]
            views[p.value] = np.ctypeslib.as_array((C.c_uint8 * (H * W)).from_address(p.value)).reshape(H, W)
        return views[p.value]

    def track_begin(self, image, window=21):
        """StripRegression(0) on the first dynamic camera image (numpy uint8 [H,W] or a CUDA tensor)."""
        ptr, stride, kind = self._image_args(image)
        self._check(lib().slx_track_begin(self._h, ptr, stride, kind, int(window)))

    def track_next(self, image):
        """One dynamic frame: strips, deltaP, U, z (x, y), deltaZ are updated in place."""
        ptr, stride, kind = self._image_args(image)
        self._check(lib().slx_track_next(self._h, ptr, stride, kind))

    def _images_args(self, images):
        if isinstance(images, np.ndarray):
            assert images.dtype == np.uint8 and images.ndim == 3 and (images.strides[2] == 1 or images.size == 0)
            return images.ctypes.data, max(images.strides[1], images.shape[2]), images.strides[0], images.shape[0], MEM_HOST
        assert images.is_cuda and images.dim() == 3 and images.stride(2) == 1
        self._borrowed.append(images)
        return images.data_ptr(), images.stride(1), images.stride(0), images.shape[0], MEM_DEVICE

    def track_next_batch(self, images, deltaz_all=None):
        """k dynamic frames in one call (slx_track_next_batch): images uint8 [k, H, W] (numpy: one transfer; CUDA tensor: borrowed);
        deltaz_all: CUDA or numpy float64 [k, H, W] receiving every frame's deltaZ, or None."""
        ptr, stride, istride, k, kind = self._images_args(images)
        dz, dz_kind = None, MEM_DEVICE
        if deltaz_all is not None:
            assert tuple(deltaz_all.shape) == (k, self.spec["height"], self.spec["width"])
            if isinstance(deltaz_all, np.ndarray):
                assert deltaz_all.dtype == np.float64 and deltaz_all.flags.c_contiguous
                dz, dz_kind = deltaz_all.ctypes.data, MEM_HOST
            else:
                assert deltaz_all.is_cuda and deltaz_all.is_contiguous()
                dz = deltaz_all.data_ptr()
        self._check(lib().slx_track_next_batch(self._h, ptr, stride, istride, k, kind, dz, dz_kind))

    def track_stage_frames(self, images):
        """One transfer of k host images (numpy uint8 [k, H, W]) into a device slab of the context; returns its device address
        (image f at + f * H * W, W bytes per row) for track_next_device."""
        ptr, stride, istride, k, kind = self._images_args(images)
        assert kind == MEM_HOST
        dev = C.c_void_p()
        self._check(lib().slx_track_stage_frames(self._h, ptr, stride, istride, k, C.byref(dev)))
        return dev.value

    def track_next_device(self, address, stride=None):
        """slx_track_next on an image in device memory given by its address (e.g. inside the slab of track_stage_frames)."""
        self._check(lib().slx_track_next(self._h, address, self.spec["width"] if stride is None else stride, MEM_DEVICE))

    def track_frames_buffer(self, k):
        """The pinned slab the next staging call / host-fed batch of up to k images copies from, as a numpy uint8 [k, H, W] view."""
        p, stride, istride = C.c_void_p(), C.c_size_t(), C.c_size_t()
        self._check(lib().slx_track_frames_buffer(self._h, int(k), C.byref(p), C.byref(stride), C.byref(istride)))
        H, W = self.spec["height"], self.spec["width"]
        assert stride.value == W and istride.value == H * W
        return np.ctypeslib.as_array((C.c_uint8 * (k * H * W)).from_address(p.value)).reshape(k, H, W)

    def get_calibration(self):
        P = (C.c_double * 12)()
        cA, cB = C.c_double(), C.c_double()
        self._check(lib().slx_get_calibration(self._h, P, C.byref(cA), C.byref(cB)))
        return np.array(P[:]).reshape(3, 4), cA.value, cB.value

    def enable_timing(self, on=True):
        self._check(lib().slx_enable_timing(self._h, 1 if on else 0))

    def last_decode_ms(self):
        ms = C.c_float()
        self._check(lib().slx_last_decode_ms(self._h, C.byref(ms)))
        return ms.value

    def debug_stamps(self, tensor):
        """tensor: CUDA int64 tensor of >= 32768 elements (or None to stop)."""
        if tensor is None:
            self._check(lib().slx_debug_stamps(self._h, None, 0))
        else:
            self._stamps = tensor
            self._check(lib().slx_debug_stamps(self._h, tensor.data_ptr(), tensor.numel()))

    def scatter_rows(self, scatters, staging, full, stream=None):
        """slx_scatter_rows: the row scatter of a staged gather group ([(src, dst, run, n_runs, src_stride, dst_stride)] from
        gather_plan_ex) from a staging buffer into the full array, both CUDA f64 tensors; asynchronous on the context's stream."""
        arr = (SlxScatter * max(len(scatters), 1))()
        for i, q in enumerate(scatters):
            arr[i] = SlxScatter(*[int(v) for v in q])
        self._check(lib().slx_scatter_rows(self._h, arr, len(scatters), staging.data_ptr(), full.data_ptr(), stream))

    def set_variant(self, v):
        self._check(lib().slx_set_variant(self._h, int(v)))

    def last_kernel(self):
        """The kernel the last decode launch of this context ran as, and how its work was cut (slx_last_kernel)."""
        buf = C.create_string_buffer(160)
        self._check(lib().slx_last_kernel(self._h, buf, 160))
        return buf.value.decode()

    def set_tuning(self, **kv):
        """Launch-geometry overrides of the fast kernel (slx_set_tuning): strip_rows, tail_pct, tail_rows, gray_plain,
        strip_waves, lds_pad_kib, plain_order; 0 = automatic."""
        for k, v in kv.items():
            self._check(lib().slx_set_tuning(self._h, TUNE_KEYS[k], int(v)))


VARIANT_AUTO, VARIANT_GENERIC, VARIANT_STRIP, VARIANT_GENERIC_FAST = range(4)
TUNE_KEYS = {"strip_rows": 0, "tail_pct": 1, "tail_rows": 2, "gray_plain": 3, "strip_waves": 4, "lds_pad_kib": 5, "plain_order": 6, "tiers": 7, "weave": 8, "stream": 9, "stream_rows": 10, "cloud_passes": 11, "cloud_spin": 12, "text_pieces": 13}


class Pipe:
    """Frame ingest pipeline over a Context (slx_pipe_*): pinned host slots, copy-in / decode / copy-out overlapped.

        pipe = Pipe(ctx, slots=3, sets_per_slot=8)
        buf = pipe.acquire()          # uint8 [sets_per_slot, planes, H, pitch], pinned: write the frames here
        pipe.submit(n_sets)
        z = pipe.collect()            # float64 [n_sets, H, W] view of the pinned result, valid until the slot is reused
    """

    def __init__(self, ctx, slots=2, sets_per_slot=1, host_result=True):
        self._ctx = ctx
        self._h = C.c_void_p()
        cfg = SlxPipeConfig(slots, sets_per_slot, 1 if host_result else 0)
        rc = lib().slx_pipe_create(ctx._h, C.byref(cfg), C.byref(self._h))
        if rc != OK:
            self._h = C.c_void_p()
            raise SlxError(rc, ctx.last_error())
        n, pitch, plane, sset = C.c_int(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        lib().slx_pipe_layout(self._h, C.byref(n), C.byref(pitch), C.byref(plane), C.byref(sset))
        self.n_planes, self.pitch, self.plane_bytes, self.set_bytes = n.value, pitch.value, plane.value, sset.value
        self.sets_per_slot, self.host_result = sets_per_slot, bool(host_result)
        self.height, self.width = ctx.spec["height"], ctx.spec["width"]

    def close(self):
        if self._h:
            lib().slx_pipe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise SlxError(rc, lib().slx_pipe_last_error(self._h).decode())

    def acquire(self):
        p = C.c_void_p()
        self._check(lib().slx_pipe_acquire(self._h, C.byref(p)))
        buf = (C.c_uint8 * (self.set_bytes * self.sets_per_slot)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.sets_per_slot, self.n_planes, self.height, self.pitch)

    def submit(self, n_sets=None):
        self._check(lib().slx_pipe_submit(self._h, self.sets_per_slot if n_sets is None else int(n_sets)))

    def collect(self, device=False):
        """float64 [n_sets, H, W] view of the pinned result (or, device=True, the device address and n_sets)."""
        h, d, n = C.c_void_p(), C.c_void_p(), C.c_int()
        self._check(lib().slx_pipe_collect(self._h, C.byref(h), C.byref(d), C.byref(n)))
        if device or not self.host_result:
            return d.value, n.value
        buf = (C.c_double * (n.value * self.height * self.width)).from_address(h.value)
        return np.frombuffer(buf, dtype=np.float64).reshape(n.value, self.height, self.width)


def comm_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 makes it; the other ranks receive it by any means)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().slx_comm_unique_id(buf, COMM_ID_BYTES)
    if rc != OK:
        raise SlxError(rc, lib().slx_comm_last_error(None).decode())
    return buf.raw


def gather_plan(shards, rank, height, width, first, count, local_plane_stride=0, root=0):
    """slx_gather_plan: [(peer, send, offset, count)] -- the messages rank `rank` posts for one group (no GPU needed)."""
    n = C.c_int(0)
    t = shard_table(shards)
    rc = lib().slx_gather_plan(t, len(shards), rank, height, width, first, count, local_plane_stride, root, None, 0, C.byref(n))
    if rc != OK:
        raise SlxError(rc, "slx_gather_plan")
    buf = (SlxMsg * max(n.value, 1))()
    rc = lib().slx_gather_plan(t, len(shards), rank, height, width, first, count, local_plane_stride, root, buf, n.value, C.byref(n))
    if rc != OK:
        raise SlxError(rc, "slx_gather_plan")
    return [(m.peer, m.send, m.offset, m.count) for m in buf[: n.value]]


def gather_plan_ex(shards, rank, height, width, first, count, local_plane_stride=0, root=0, shape=GATHER_IN_PLACE):
    """slx_gather_plan_ex: (messages, scatters, staging_doubles) of one group for either gather shape (no GPU needed).
    messages: [(peer, send, offset, count)] with send 0 = receive into full, 1 = send from local, 2 = receive into the staging slot;
    scatters: [(src, dst, run, n_runs, src_stride, dst_stride)] -- how the staging slot then goes to the full array."""
    shape = GATHER_SHAPES.get(shape, shape)
    t = shard_table(shards)
    n, ns, st = C.c_int(0), C.c_int(0), C.c_ulonglong(0)
    args = (t, len(shards), rank, height, width, first, count, local_plane_stride, root, shape)
    rc = lib().slx_gather_plan_ex(*args, None, 0, C.byref(n), None, 0, C.byref(ns), C.byref(st))
    if rc != OK:
        raise SlxError(rc, "slx_gather_plan_ex")
    buf, sbuf = (SlxMsg * max(n.value, 1))(), (SlxScatter * max(ns.value, 1))()
    rc = lib().slx_gather_plan_ex(*args, buf, n.value, C.byref(n), sbuf, ns.value, C.byref(ns), C.byref(st))
    if rc != OK:
        raise SlxError(rc, "slx_gather_plan_ex")
    return ([(m.peer, m.send, m.offset, m.count) for m in buf[: n.value]],
            [(q.src, q.dst, q.run, q.n_runs, q.src_stride, q.dst_stride) for q in sbuf[: ns.value]], st.value)


REFERENCE_DEFAULT_NAMES = ("PROJECTOR_RESLINE", "PROJECTOR_RESROW", "CAMERA_RESLINE", "CAMERA_RESROW", "GRAY_V_NUMDIGIT", "PHASE_NUMDIGIT",
                           "FOV_MIN_DISTANCE", "FOV_MAX_DISTANCE", "RECO_WINDOW_SIZE", "DYNAFRAME_MAXNUM")


def reference_defaults():
    """{name: value} of the reference's compiled-in configuration as the C++ mirror classes default to it (slx_reference_defaults)."""
    n = C.c_int(0)
    lib().slx_reference_defaults(None, 0, C.byref(n))
    v = (C.c_int * n.value)()
    rc = lib().slx_reference_defaults(v, n.value, C.byref(n))
    if rc != OK or n.value != len(REFERENCE_DEFAULT_NAMES):
        raise SlxError(rc, "slx_reference_defaults")
    return dict(zip(REFERENCE_DEFAULT_NAMES, [int(x) for x in v]))


def shard_table(shards):
    """[(set0, n_sets, row0, rows), ...] per rank -> the C array slx_gather_depth takes."""
    arr = (SlxShard * len(shards))()
    for i, (set0, n, row0, rows) in enumerate(shards):
        arr[i] = SlxShard(int(set0), int(n), int(row0), int(rows))
    return arr


class Comm:
    """The RCCL communicator of the depth-map gather (slx_comm_*), one per rank, bound to a Context."""

    def __init__(self, ctx, unique_id, world, rank):
        self._ctx = ctx
        self._h = C.c_void_p()
        rc = lib().slx_comm_create(ctx._h, unique_id, len(unique_id), int(world), int(rank), C.byref(self._h))
        if rc != OK:
            self._h = C.c_void_p()
            raise SlxError(rc, lib().slx_comm_last_error(None).decode())
        try:
            self.world, self.rank = self.info()
        except Exception:
            self.close()                                             # the communicator exists: do not leak it with the exception
            raise

    def info(self):
        """(ranks, this rank) as RCCL reports them for the communicator the gather runs on (ncclCommCount / ncclCommUserRank)."""
        w, r = C.c_int(), C.c_int()
        rc = lib().slx_comm_info(self._h, C.byref(w), C.byref(r))
        if rc != OK:
            raise SlxError(rc, "slx_comm_info failed")
        return w.value, r.value

    def close(self):
        if self._h:
            lib().slx_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise SlxError(rc, lib().slx_comm_last_error(self._h).decode())

    def synchronize(self):
        self._check(lib().slx_comm_synchronize(self._h))

    def set_gather_shape(self, shape):
        """"in_place" (one message per (peer, frame-set), landing in place) or "staged" (one message per (peer, chunk) into a staging
        slot of the root + a row-scatter kernel): slx_comm_set_gather_shape.  Every rank must set the same shape."""
        self._check(lib().slx_comm_set_gather_shape(self._h, GATHER_SHAPES.get(shape, shape)))

    def gather_depth(self, shards, height, width, local, full, root=0, local_plane_stride=0, stream=None):
        """shards: [(set0, n_sets, row0, rows)] per rank; local / full: CUDA float64 tensors (full may be None on ranks that
        do not receive).  Asynchronous on the comm's gather stream (or `stream`)."""
        self._check(lib().slx_gather_depth(self._h, shard_table(shards), int(height), int(width), None if local is None else local.data_ptr(),
                                           int(local_plane_stride), None if full is None else full.data_ptr(), int(root), stream))

    def decode_gather(self, shards, full_height, chunk_sets, phase, gray, scratch, full, root=0, stream=None, row_stride=None, ctx=None):
        """slx_decode_gather: this rank's shard decoded chunk by chunk, every chunk gathered while the next decodes.
        ctx: the context that decodes (default: the one the communicator was created with; any context of that device will do)."""
        def base(t):
            if t is None:
                return None, 0
            assert t.is_cuda and t.stride(-1) == 1
            return t.data_ptr(), t.stride(0) * t.element_size()
        pb, ps = base(phase)
        gb, gs = base(gray)
        ref = phase if phase is not None else gray
        rs = ref.stride(-2) if row_stride is None else row_stride
        self._check(lib().slx_decode_gather(self._h, (ctx or self._ctx)._h, shard_table(shards), int(full_height), int(chunk_sets), pb, ps, gb, gs, rs,
                                            None if scratch is None else scratch.data_ptr(), None if full is None else full.data_ptr(),
                                            int(root), stream))


def _read_gray_file(fn, path):
    r, c = C.c_int(), C.c_int()
    rc = fn(path.encode(), None, 0, C.byref(r), C.byref(c))
    if rc != OK:
        raise SlxError(rc, "cannot read %s" % path)
    a = np.empty((r.value, c.value), dtype=np.uint8)
    rc = fn(path.encode(), a.ctypes.data, a.size, C.byref(r), C.byref(c))
    if rc != OK:
        raise SlxError(rc, "cannot read %s" % path)
    return a


def read_bmp_gray(path):
    """uint8 [rows, cols] of an uncompressed BMP, converted like imread(..., CV_LOAD_IMAGE_GRAYSCALE)."""
    return _read_gray_file(lib().slx_read_bmp_gray, path)


def read_pgm_gray(path):
    """uint8 [rows, cols] of a binary PGM (P5, maxval <= 255)."""
    return _read_gray_file(lib().slx_read_pgm_gray, path)


TEXT_LIBSTDCXX, TEXT_MSVC2013 = 0, 1      # enum slx_text_dialect


def write_point_cloud_text(path, xyz, dialect=TEXT_LIBSTDCXX):
    """The text file CCalculation::Result writes (R/CCalculation.cpp:323-357): "x y z" per line, numbers as `ostream << double`
    prints them.  xyz: float64 [n, 3] (what Context.point_cloud returns).  dialect: TEXT_LIBSTDCXX ("5e-05", LF) or TEXT_MSVC2013
    (the reference as built: "5e-005", CR LF)."""
    a = np.ascontiguousarray(xyz, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("xyz must be [n, 3]")
    rc = lib().slx_write_point_cloud_text_ex(os.fsencode(path), a.ctypes.data if a.size else None, a.shape[0], int(dialect))
    if rc != OK:
        raise SlxError(rc, "cannot write %s" % path)


def read_calibration_yaml(path):
    """{'cam','pro','rot','trans'} from the cv::FileStorage YAML CCalculation::Init reads."""
    bufs = [(C.c_double * n)() for n in (9, 9, 9, 3)]
    rc = lib().slx_read_calibration_yaml(path.encode(), *bufs)
    if rc != OK:
        raise SlxError(rc, "cannot read %s" % path)
    return {k: list(b) for k, b in zip(("cam", "pro", "rot", "trans"), bufs)}


def decode_frameset(spec, phase=None, gray=None, want=("z",), device=-1, variant=0, tune=None, info=None):
    """Convenience: one frame-set from host arrays, outputs as numpy arrays.  tune: slx_set_tuning overrides; info: a dict that
    receives "kernel" = slx_last_kernel's text for the launch."""
    aux = [w for w in want if w != "z" or spec["mode"] < MODE_GRAY_PHASE]
    primary = {MODE_PHASE_ONLY: "pix", MODE_GRAY_ONLY: "gray"}.get(spec["mode"])
    aux = [w for w in aux if w != primary]
    with Context(spec, device=device, aux=aux) as ctx:
        ctx.set_variant(variant)
        if tune:
            ctx.set_tuning(**tune)
        ctx.set_frames(phase, gray)
        ctx.decode()
        if info is not None:
            info["kernel"] = ctx.last_kernel()
        return {w: ctx.get_output(w) for w in want}
