#!/usr/bin/env python3
"""GPU: ONE frame-set per launch, back to back (the reference's own use: a depth map per capture) -- plain launches on the context's stream,
plain launches on a caller's stream, and the same launches captured once into a hipGraph and replayed (torch.cuda.CUDAGraph: stream capture
of slx_decode_batch_ex on the caller's stream).  12 distinct frame-sets in rotation (every launch reads HBM).  Usage: tools/graph_single_set.py [--config C4]"""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C4")
ap.add_argument("--rotate", type=int, default=12)
ap.add_argument("--sets", type=int, default=1, help="frame-sets per launch (a batch: from 5 on the planner takes the stream kernel for C4, also inside a graph since round 6)")
ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE")
a = ap.parse_args()
spec = synth.make_spec(a.config)
H, W = spec["height"], spec["width"]
R, S = a.rotate, a.sets
rng = np.random.default_rng(1)
n_ph = spec["n_freq"] * spec["n_steps"] if spec["mode"] != synth.MODE_GRAY_ONLY else 0
n_gr = 2 * spec["gray_bits"]
ph = torch.from_numpy(rng.integers(0, 256, (R * S, max(n_ph, 1), H, W), dtype=np.uint8)).cuda() if n_ph else None
gr = torch.from_numpy(rng.integers(0, 256, (R * S, n_gr, H, W), dtype=np.uint8)).cuda() if n_gr else None
z = torch.empty((R * S, H, W), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
out = {"config": a.config, "rotate": R, "sets_per_launch": S, "tune": a.tune}
bytes_per = S * H * W * synth.algorithmic_bytes_per_pixel(spec)
with api.Context(spec) as ctx:
    if a.tune:
        ctx.set_tuning(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.tune})

    def launch(r, stream=None):
        ctx.decode_batch_ex(S, None if ph is None else ph[r * S:(r + 1) * S], None if gr is None else gr[r * S:(r + 1) * S], z=z[r * S:(r + 1) * S], stream=stream)
    for r in range(R):
        launch(r)
    ctx.synchronize()
    ref = z.clone()

    def timed(fn, n):
        fn(); torch.cuda.synchronize(); ctx.synchronize()
        best = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            ctx.synchronize(); torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / n)
        return sorted(best)[2]

    def plain():
        for r in range(R):
            launch(r)
    t = timed(plain, 40) / R
    out["plain_own_stream_us"] = t * 1e6
    s = torch.cuda.Stream()

    def plain_caller():
        for r in range(R):
            launch(r, stream=s.cuda_stream)
    t = timed(plain_caller, 40) / R
    out["plain_caller_stream_us"] = t * 1e6
    try:
        g = torch.cuda.CUDAGraph()
        z.zero_()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for r in range(R):
                launch(r, stream=s.cuda_stream)
        g.replay(); torch.cuda.synchronize()
        out["graph_equals_plain"] = bool(torch.equal(z, ref))
        out["captured_kernel"] = ctx.last_kernel()
        t = timed(g.replay, 40) / R
        out["graph_replay_us"] = t * 1e6
    except Exception as e:
        out["graph_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
for k in list(out):
    if k.endswith("_us"):
        out[k.replace("_us", "_frac_of_hbm_peak")] = bytes_per / (out[k] * 1e-6) / 8e12
print(json.dumps(out))
