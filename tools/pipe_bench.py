#!/usr/bin/env python3
"""PCIe-inclusive throughput of the frame ingest pipeline (slx_pipe_*) on the GPU box.

Frame-sets start in pinned HOST memory and the depth maps end in pinned HOST memory: copy-in, decode and copy-out of
consecutive slots overlap on three HIP streams.  The pinned inputs are filled once (a camera SDK would DMA into them);
the timed loop is acquire -> submit -> collect with the slots kept full.  This number is NOT bench.py's `value`
(which starts with the inputs resident in HBM); DESIGN.md quotes it beside it.

Usage: tools/pipe_bench.py [--config C4] [--slots 3] [--sets-per-slot 8] [--submits 40] [--device-result]
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C4")
ap.add_argument("--slots", type=int, default=3)
ap.add_argument("--sets-per-slot", type=int, default=8)
ap.add_argument("--submits", type=int, default=40)
ap.add_argument("--device-result", action="store_true", help="leave the depth on the device (copy-in + decode only)")
a = ap.parse_args()

spec = synth.make_spec(a.config)
H, W = spec["height"], spec["width"]
ph, gr, _ = synth.render(spec, "sphere", seed=5, noise_sigma=2.0)
planes = np.concatenate([x for x in (ph, gr) if x is not None])
with api.Context(spec) as ctx:
    pipe = api.Pipe(ctx, slots=a.slots, sets_per_slot=a.sets_per_slot, host_result=not a.device_result)
    # fill every slot once: cycle through them with tiny submits
    for _ in range(a.slots):
        buf = pipe.acquire()
        buf[:, :, :, :W] = planes[None]
        pipe.submit(a.sets_per_slot)
    for _ in range(a.slots):
        pipe.collect()
    in_flight = 0
    for it in range(a.slots):                       # warm-up round, slots full
        pipe.acquire()
        pipe.submit(a.sets_per_slot)
        in_flight += 1
    t0 = time.perf_counter()
    for it in range(a.submits):
        pipe.collect()
        pipe.acquire()
        pipe.submit(a.sets_per_slot)
    t1 = time.perf_counter()
    while in_flight:
        pipe.collect()
        in_flight -= 1
    pipe.close()
sets = a.submits * a.sets_per_slot
dt = t1 - t0
in_bytes = pipe.set_bytes * sets
out_bytes = 0 if a.device_result else H * W * 8 * sets
print(json.dumps({
    "metric": "frame-sets/s through the ingest pipeline (host -> depth in host memory)" if not a.device_result
              else "frame-sets/s through the ingest pipeline (host -> depth on the device)",
    "config": a.config, "slots": a.slots, "sets_per_slot": a.sets_per_slot, "submits": a.submits,
    "value": sets / dt, "ms_per_frameset": dt / sets * 1e3,
    "h2d_GBps": in_bytes / dt / 1e9, "d2h_GBps": out_bytes / dt / 1e9,
}))
