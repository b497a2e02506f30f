#!/usr/bin/env python3
"""Same-box A/B of the launch with the optional planes (x, y, U, k beside z: slx_decode_batch_ex) on the strip kernel (stream=1) and on
the stream kernel's optional-plane instantiation (stream=2), interleaved round-robin, medians.  Planes are allocated one allocation per
plane before anything else.  Usage: tools/aux_ab.py [--config C4] [--sets 16] [--rounds 7]"""
import argparse, importlib, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C4")
ap.add_argument("--sets", type=int, default=16)
ap.add_argument("--rounds", type=int, default=7)
a = ap.parse_args()
spec = synth.make_spec(a.config)
H, W, F, n = spec["height"], spec["width"], spec["n_freq"], a.sets
outs = {w: torch.empty((n, H, W), dtype=torch.float64, device="cuda") for w in ("z", "x", "y", "U")}
outs["k"] = torch.empty((n, F - 1, H, W), dtype=torch.int32, device="cuda")
phase = torch.randint(0, 256, (n, 4 * F, H, W), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
arms = {}
for name, tune in (("strip  (stream=1)", dict(stream=1)), ("stream (stream=2)", dict(stream=2))):
    c = api.Context(spec)
    c.set_tuning(**tune)
    arms[name] = (c, torch.cuda.ExternalStream(c.stream_handle()), [])
def run(c, st, k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(k):
        c.decode_batch_ex(n, phase, None, **outs)
    e1.record(st)
    c.synchronize()
    return e0.elapsed_time(e1) * 1e3 / k
for c, st, _ in arms.values():
    run(c, st, 30)
for _ in range(a.rounds):
    for c, st, res in arms.values():
        res.append(run(c, st, 30))
bytes_ = n * H * W * (4 * F + 8 + 24 + 4 * (F - 1))
for name, (c, st, res) in arms.items():
    m = statistics.median(res)
    print("%-20s %-60s median %7.1f us  min %7.1f  -> %5.2f TB/s (%.1f %% of 8 TB/s)" % (name, c.last_kernel()[:60], m, min(res), bytes_ / m / 1e6, bytes_ / m / 1e6 / 8 * 100))
