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
a = ap.parse_args()
spec = synth.make_spec(a.config)
H, W = spec["height"], spec["width"]
R = a.rotate
rng = np.random.default_rng(1)
n_ph = spec["n_freq"] * spec["n_steps"] if spec["mode"] != synth.MODE_GRAY_ONLY else 0
n_gr = 2 * spec["gray_bits"]
ph = torch.from_numpy(rng.integers(0, 256, (R, max(n_ph, 1), H, W), dtype=np.uint8)).cuda() if n_ph else None
gr = torch.from_numpy(rng.integers(0, 256, (R, n_gr, H, W), dtype=np.uint8)).cuda() if n_gr else None
z = torch.empty((R, H, W), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
out = {"config": a.config, "rotate": R}
bytes_per = H * W * synth.algorithmic_bytes_per_pixel(spec)
with api.Context(spec) as ctx:
    def launch(r, stream=None):
        ctx.decode_batch_ex(1, None if ph is None else ph[r:r + 1], None if gr is None else gr[r:r + 1], z=z[r:r + 1], stream=stream)
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
        t = timed(g.replay, 40) / R
        out["graph_replay_us"] = t * 1e6
    except Exception as e:
        out["graph_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
for k in list(out):
    if k.endswith("_us"):
        out[k.replace("_us", "_frac_of_hbm_peak")] = bytes_per / (out[k] * 1e-6) / 8e12
print(json.dumps(out))
