#!/usr/bin/env python3
"""GPU: a dynamic frame of the reference's loop in parts at 1280 x 1024 -- read the BMP (slx_read_bmp_gray), slx_track_next + wait, slx_get_point_cloud_text,
write the 30 MB file -- ms per frame over 30 frames, twice (the second pass overwrites the first one's files).  Usage: tools/frame_parts.py"""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from dynaframe_files import write_bmp
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
L = api.lib(); libc = C.CDLL(None)
spec = synth.make_spec("REF"); H, W = spec["height"], spec["width"]
ph, gr, _ = synth.render(spec, "sphere", seed=5, noise_sigma=2.0)
rng = np.random.default_rng(8)
u = np.arange(W)[None, :] + 0.03 * np.arange(H)[:, None]
d = "/tmp/slx_loop_parts"; os.makedirs(d, exist_ok=True)
N = 30
for f in range(N + 1):
    img = 128 + 100 * np.sign(np.sin(2 * np.pi * (u + 1.7 * f) / 14.0)) + rng.normal(0, 6, (H, W))
    write_bmp(os.path.join(d, "dyna%d.bmp" % f), np.clip(img, 0, 255).astype(np.uint8), bits=8)
acc = {"read_bmp": 0.0, "track": 0.0, "text": 0.0, "write": 0.0}
buf = np.empty((H, W), dtype=np.uint8)
with api.Context(spec, aux=("U",)) as ctx:
    ctx.set_frames(ph, gr); ctx.decode()
    r, c = C.c_int(), C.c_int()
    assert L.slx_read_bmp_gray(os.path.join(d, "dyna0.bmp").encode(), buf.ctypes.data, buf.size, C.byref(r), C.byref(c)) == 0
    ctx.track_begin(buf)
    for rep in range(2):
        for k in acc: acc[k] = 0.0
        for f in range(1, N + 1):
            t0 = time.perf_counter()
            assert L.slx_read_bmp_gray(os.path.join(d, "dyna%d.bmp" % f).encode(), buf.ctypes.data, buf.size, C.byref(r), C.byref(c)) == 0
            t1 = time.perf_counter()
            ctx.track_next(buf); ctx.synchronize()
            t2 = time.perf_counter()
            tp, nb, npts = C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
            assert L.slx_get_point_cloud_text(ctx._h, C.byref(tp), C.byref(nb), C.byref(npts)) == 0
            t3 = time.perf_counter()
            fd = os.open(os.path.join(d, "cloud%d.txt" % f), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            done = 0
            while done < nb.value: done += libc.write(fd, C.c_void_p(tp.value + done), C.c_size_t(nb.value - done))
            os.close(fd)
            t4 = time.perf_counter()
            acc["read_bmp"] += t1 - t0; acc["track"] += t2 - t1; acc["text"] += t3 - t2; acc["write"] += t4 - t3
        print("pass", rep, {k: round(v / N * 1e3, 2) for k, v in acc.items()}, "ms per frame; text bytes", nb.value, flush=True)
import shutil; shutil.rmtree(d)
