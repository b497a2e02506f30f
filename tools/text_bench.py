#!/usr/bin/env python3
"""GPU: the point-cloud TEXT of CCalculation::Result (R/CCalculation.cpp:323-357) for one frame, two ways -- the cloud to pinned host memory and
the host formatter (slx_get_point_cloud_view + slx_write_point_cloud_text, 16 threads), or formatted on the device and written as it arrives
(slx_get_point_cloud_text + one fwrite) -- to /dev/null (formatting + transfers alone) and to a file.  Usage: tools/text_bench.py [--config C4]"""
import argparse, ctypes as C, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C4")
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--dir", default="/tmp")
ap.add_argument("--lib", default="", help="another build of libslx.so")
a = ap.parse_args()
if a.lib:
    api.LIB_PATH = os.path.join(ROOT, a.lib)
spec = synth.make_spec(a.config)
ph, gr, _ = synth.render(spec, "sphere", seed=9, noise_sigma=1.0)
L = api.lib()
libc = C.CDLL(None)
out = {"config": a.config, "pixels": spec["height"] * spec["width"]}
with api.Context(spec) as ctx:
    ctx.set_frames(phase=ph, gray=gr)
    ctx.decode()
    ctx.synchronize()
    for target in ("/dev/null", os.path.join(a.dir, "slx_text_bench.txt")):
        host, dev, dev_only = [], [], []
        for rep in range(a.reps):
            t0 = time.perf_counter()
            p, n = C.c_void_p(), C.c_size_t(0)
            assert L.slx_get_point_cloud_view(ctx._h, C.byref(p), C.byref(n)) == 0
            assert L.slx_write_point_cloud_text(target.encode(), p, n) == 0
            host.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            tp, nb, npts = C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
            assert L.slx_get_point_cloud_text(ctx._h, C.byref(tp), C.byref(nb), C.byref(npts)) == 0
            t1 = time.perf_counter()
            fd = os.open(target, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            done = 0
            while done < nb.value:
                done += libc.write(fd, C.c_void_p(tp.value + done), C.c_size_t(nb.value - done))
            os.close(fd)
            dev.append(time.perf_counter() - t0)
            dev_only.append(t1 - t0)
        key = "dev_null" if target == "/dev/null" else "file"
        out[key] = {"host_formatter_ms": [round(t * 1e3, 2) for t in host], "device_formatter_ms": [round(t * 1e3, 2) for t in dev],
                    "device_formatter_without_the_write_ms": [round(t * 1e3, 2) for t in dev_only]}
        out["points"], out["text_bytes"] = npts.value, nb.value
        if target != "/dev/null":
            os.remove(target)
    # the call alone, by the number of pieces the text is formatted and copied in (1 = cloud, text, copy one after the other), beside the
    # time the same number of bytes takes from device to pinned host memory on a copy engine
    import statistics
    sweep = {}
    for pieces in (1, 2, 4, 8, 16, 0):
        ctx.set_tuning(text_pieces=pieces)
        ts = []
        for rep in range(24):
            t0 = time.perf_counter()
            tp, nb, npts = C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
            assert L.slx_get_point_cloud_text(ctx._h, C.byref(tp), C.byref(nb), C.byref(npts)) == 0
            ts.append(time.perf_counter() - t0)
        sweep["default (2)" if pieces == 0 else str(pieces)] = {"median_ms": round(statistics.median(ts[4:]) * 1e3, 4), "min_ms": round(min(ts[4:]) * 1e3, 4)}
    out["call_ms_by_pieces"] = sweep
    src = torch.empty(out["text_bytes"], dtype=torch.uint8, device="cuda")
    dst = torch.empty(out["text_bytes"], dtype=torch.uint8).pin_memory()
    ts = []
    for rep in range(24):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out["pcie_copy_of_the_text_ms"] = {"median": round(statistics.median(ts[4:]) * 1e3, 4), "min": round(min(ts[4:]) * 1e3, 4)}
print(json.dumps(out))
