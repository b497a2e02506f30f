#!/usr/bin/env python3
"""Diagnostic (GPU box): what lies between two back-to-back launches of the strip kernel.  Two contexts with a stamp buffer
each decode alternately on one stream; s_memrealtime is one 100 MHz counter for the whole chip, so the last wave's end stamp of
launch i and the first wave's start stamp of launch i+1 give the idle gap between them."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
n_sets = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spec = synth.make_spec(cfg)
H, W = spec["height"], spec["width"]
n_phase, n_gray = synth.n_planes(spec)
phase = torch.randint(0, 256, (n_sets, n_phase, H, W), dtype=torch.uint8, device="cuda")
gray = torch.randint(0, 256, (n_sets, n_gray, H, W), dtype=torch.uint8, device="cuda") if n_gray else None
z = torch.empty((n_sets, H, W), dtype=torch.float64, device="cuda")
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctxs, stamps = [], []
for i in range(2):
    c = api.Context(spec); c.set_variant(2)
    st = torch.zeros(4 * 65536, dtype=torch.int64, device="cuda")
    ctxs.append(c); stamps.append(st)
for _ in range(150):
    for c in ctxs:
        c.decode_batch(n_sets, phase, gray, z, stream=s.cuda_stream)
torch.cuda.synchronize()
for c, st in zip(ctxs, stamps):
    c.debug_stamps(st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s)
for _ in range(20):
    for c in ctxs:
        c.decode_batch(n_sets, phase, gray, z, stream=s.cuda_stream)
e1.record(s)
torch.cuda.synchronize()
print("event-timed launch (stamps on): %.1f us" % (e0.elapsed_time(e1) * 1000 / 40))
w = [st.cpu().numpy().reshape(-1, 4) for st in stamps]
w = [x[x[:, 1] > 0] for x in w]
a0, a1 = w[0][:, 2].min(), w[0][:, 3].max()       # launch 39 (context 0 ran second to last)
b0, b1 = w[1][:, 2].min(), w[1][:, 3].max()       # launch 40
print("launch A: first start .. last end = %.1f us ; launch B: %.1f us" % ((a1 - a0) / 100.0, (b1 - b0) / 100.0))
print("gap: last wave end of A -> first wave start of B = %.2f us" % ((b0 - a1) / 100.0))
print("period A start -> B start = %.1f us" % ((b0 - a0) / 100.0))
for name, x, t0, t1 in (("A", w[0], a0, a1), ("B", w[1], b0, b1)):
    ts = np.linspace(t0, t1, 40)
    print(name, "alive:", [int(((x[:, 2] <= t) & (x[:, 3] > t)).sum()) for t in ts])
