#!/usr/bin/env python3
"""Three clocks for the launches that last 10-35 us (GPU box): the dynamic frame (slx_track_fused_kernel<10>), the fused point cloud
(slx_cloud_fused_kernel) and ONE frame-set of configuration 4 (slx_strip_kernel<3, 3, 0, 4, false>), each at 1920 x 1200:
  wall     host clock over N back-to-back calls, as bench.py's around_the_path / other_configs entries take it (launch gaps included;
           the cloud call also holds its wait for the count)
  events   HIP events on the launch stream around the same N calls
  stamps   s_memrealtime (100 MHz) in the kernel itself: first workgroup's start to last workgroup's end of ONE launch, median of M launches
and, when run under `rocprofv3 --kernel-trace` (tools/profile_short.sh), the profiler's interval per dispatch, from which the collector
takes the average duration AND start-to-start / end-to-end periods of consecutive dispatches (intervals that overlap say the
profiler's begin stamp is taken before the previous kernel has drained).
Usage: tools/short_kernels.py [--n 400] [--m 60] [--only tracker,cloud,c4x1] [--no-stamps]"""
import argparse
import ctypes as C
import importlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400)
ap.add_argument("--m", type=int, default=60)
ap.add_argument("--only", default="tracker,cloud,c4x1")
ap.add_argument("--no-stamps", action="store_true", help="skip the in-kernel clock (the runs under the profiler: its intervals are wanted for the plain kernels)")
a = ap.parse_args()
only = set(a.only.split(","))
dev = torch.device("cuda")
spec = synth.make_spec("C4")
H, W = spec["height"], spec["width"]
out = {"n_back_to_back": a.n, "m_stamped_launches": a.m}
L = api.lib()


def settle(fn, sync, seconds=0.15):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        sync()


def clocks(fn, sync, stream, ctx, n_wg_hint):
    """wall / events over a.n calls; stamps over a.m single launches."""
    settle(fn, sync)
    res = {}
    blocks_wall, blocks_ev = [], []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync()
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(a.n):
            fn()
        e1.record(stream)
        sync()
        e1.synchronize()
        blocks_wall.append((time.perf_counter() - t0) / a.n * 1e6)
        blocks_ev.append(e0.elapsed_time(e1) / a.n * 1e3)
    res["wall_us"] = sorted(blocks_wall)[2]
    res["events_us"] = sorted(blocks_ev)[2]
    res["wall_us_blocks"], res["events_us_blocks"] = blocks_wall, blocks_ev
    if not a.no_stamps:
        st = torch.zeros(4 * 16384, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ctx.debug_stamps(st)
        spans, skews, alive = [], [], []
        for k in range(a.m + 5):
            for _ in range(8):
                fn()                                   # the launch measured follows others back to back, as in the loops above
            sync()
            st.zero_()
            torch.cuda.synchronize()
            fn()
            sync()
            torch.cuda.synchronize()
            w = st.cpu().numpy().reshape(-1, 4)
            w = w[(w[:, 3] > 0) & (w[:, 2] > 0)]
            if k >= 5 and len(w):
                spans.append((w[:, 3].max() - w[:, 2].min()) / 100.0)
                skews.append((w[:, 2].max() - w[:, 2].min()) / 100.0)
                alive.append(len(w))
        ctx.debug_stamps(None)
        res["stamps_us"] = statistics.median(spans)
        res["stamps_us_min_max"] = [min(spans), max(spans)]
        res["stamps_start_skew_us"] = statistics.median(skews)
        res["stamped_workgroups_or_items"] = int(statistics.median(alive))
        res["stamps_note"] = "the launch measured is the last of 9 back-to-back calls followed by a wait; s_memrealtime ticks at 100 MHz (10 ns)"
    return res


if "tracker" in only or "cloud" in only:
    ph, gr, _ = synth.render(spec, "sphere", seed=9, noise_sigma=1.0)
    u = torch.arange(W, device=dev)[None, :] + 0.02 * torch.arange(H, device=dev)[:, None]
    gen = torch.Generator(device=dev).manual_seed(3)
    imgs = [(128 + 100 * torch.sign(torch.sin(2 * np.pi * (u + 1.7 * f) / 14.0)) + 6 * torch.randn((H, W), device=dev, generator=gen)).clamp(0, 255).to(torch.uint8)
            for f in range(4)]
    xyz = torch.empty((H * W, 3), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    with api.Context(spec, aux=("U",)) as ctx:
        stream = torch.cuda.ExternalStream(ctx.stream_handle(), device=dev)
        ctx.set_frames(ph, gr)
        ctx.decode()
        ctx.track_begin(imgs[0])
        turn = [0]

        def track():
            turn[0] += 1
            ctx.track_next(imgs[turn[0] % 4])
        if "tracker" in only:
            r = clocks(track, ctx.synchronize, stream, ctx, 1200)
            r.update({"kernel": "slx_track_fused_kernel<10>", "algorithmic_bytes": 77 * H * W, "what": "slx_track_next, image resident in HBM, 1920x1200, window 21"})
            out["tracker"] = r
        if "cloud" in only:
            n = C.c_size_t(0)

            def cloud():
                assert L.slx_get_point_cloud(ctx._h, xyz.data_ptr(), H * W, C.byref(n), api.MEM_DEVICE) == 0
            r = clocks(cloud, lambda: None, stream, ctx, 600)       # (the call returns with its stream drained)
            r.update({"kernel": "slx_cloud_fused_kernel", "points": int(n.value), "algorithmic_bytes": 8 * H * W + 24 * int(n.value),
                      "what": "slx_get_point_cloud into device memory; wall and events include the host's wait for the count (one call = launch + hipStreamSynchronize)"})
            out["cloud"] = r
if "c4x1" in only:
    ROT = 12
    sys.path.insert(0, ROOT)
    import bench
    phase, _ = bench.make_batch(torch, synth, spec, ROT, dev, seed=0x5EED + 99)
    z = torch.empty((ROT, H, W), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        stream = torch.cuda.ExternalStream(ctx.stream_handle(), device=dev)
        turn = [0]

        def one():
            r = turn[0] % ROT
            turn[0] += 1
            ctx.decode_batch_ex(1, phase[r:r + 1], None, z=z[r:r + 1])
        r = clocks(one, ctx.synchronize, stream, ctx, 3000)
        r.update({"kernel": ctx.last_kernel(), "algorithmic_bytes": 20 * H * W, "what": "ONE frame-set of configuration 4 per launch, rotating over 12 distinct frame-sets (bench.py's C4x1)"})
        out["c4x1"] = r
for k, v in out.items():
    if isinstance(v, dict) and "algorithmic_bytes" in v:
        for clock in ("wall_us", "events_us", "stamps_us"):
            if clock in v:
                v["frac_of_8TBps_by_" + clock[:-3]] = v["algorithmic_bytes"] / v[clock] / 1e6 / 8.0
print(json.dumps(out))
