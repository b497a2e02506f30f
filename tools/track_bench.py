#!/usr/bin/env python3
"""Time of one dynamic frame (slx_track_next: StripRegression + FillOtherDeltaProU + FillCoordinate + deltaZ) on the
GPU box, camera image resident in HBM.  Usage: tools/track_bench.py [--size 1920x1200] [--frames 60]"""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1200")
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--window", type=int, default=21)
ap.add_argument("--host", action="store_true", help="feed the camera images from host memory (numpy): the pinned double-buffered path of slx_track_next")
ap.add_argument("--in-place", action="store_true", help="with --host: write each image into slx_track_image_buffer (the copy into the pinned slot is the producer's, as a camera SDK's would be) -- the timed loop then holds no host copy")
ap.add_argument("--batch", type=int, default=0, help="with --host: k images per transfer (slx_track_next_batch); with --in-place the producer writes into slx_track_frames_buffer")
a = ap.parse_args()
W, H = (int(v) for v in a.size.split("x"))
spec = dict(synth.make_spec("REF"))
spec["width"], spec["height"] = W, H
spec["calib"] = synth.scaled_calibration(W, H, spec["proj_width"])
ph, gr, _ = synth.render(spec, "sphere", seed=9, noise_sigma=1.0)
imgs = [torch.randint(0, 256, (H, W), dtype=torch.uint8, device="cuda") for _ in range(4)]
if a.host:
    imgs = [i.cpu().numpy() for i in imgs]
with api.Context(spec, aux=("U", "x", "y")) as ctx:
    ctx.set_frames(phase=ph, gray=gr)
    ctx.decode()
    ctx.synchronize()
    ctx.track_begin(imgs[0], window=a.window)
    for i in range(10):
        ctx.track_next(imgs[i % 4])
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    t_buf = t_next = 0.0
    import numpy as np
    if a.batch:
        assert a.host, "--batch feeds host images"
        k = a.batch
        stack = np.stack([imgs[i % 4] for i in range(k)])
        for _ in range(3):
            ctx.track_next_batch(stack)
        ctx.synchronize()
        t0 = time.perf_counter()
        a.frames = max(1, a.frames // k) * k
    for i in range(0 if not a.batch else a.frames // a.batch):
        if a.in_place:
            ctx.track_next_batch(ctx.track_frames_buffer(a.batch))   # contents as the producer left them
        else:
            ctx.track_next_batch(stack)
    for i in range(a.frames if not a.batch else 0):
        if a.host and a.in_place:
            ta = time.perf_counter()
            b = ctx.track_image_buffer()                             # contents as the producer left them
            tb = time.perf_counter()
            ctx.track_next(b)
            t_buf += tb - ta
            t_next += time.perf_counter() - tb
        else:
            ctx.track_next(imgs[i % 4])
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / a.frames
# algorithmic bytes per pixel of one frame: image 1; strips W,B written 8; previous strips read 8; U 8 read + 8 written;
# deltaP 4 written; z 8 read; z/x/y 24 written; deltaZ 8 written.  (The unblurred deltaP stays in LDS in the one-launch
# kernel of the 21-pixel window; other windows take two launches and move 8 more bytes per pixel.)
bytes_px = 1 + 8 + 8 + 8 + 8 + 4 + 8 + 24 + 8
print(json.dumps({"metric": "dynamic frames/s (slx_track_next)", "size": a.size, "window": a.window, "batch": a.batch or None, "images": ("host, written in place into the pinned double buffer" if a.in_place else "host (pinned double buffer)") if a.host else "device", "frames": a.frames,
                  "value": 1.0 / dt, "us_per_frame": dt * 1e6, "algorithmic_bytes_per_pixel": bytes_px,
                  "achieved_GBps": bytes_px * W * H / dt / 1e9,
                  **({"host_us_in_image_buffer": t_buf / a.frames * 1e6, "host_us_in_track_next": t_next / a.frames * 1e6} if a.host and a.in_place else {})}))
