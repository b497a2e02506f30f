#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants / tuning knobs on the bench workload (GPU box).
Usage: tools/ab.py "1" "2" "2:strip_rows=8" "2@tmp_ab/libslx_base.so" ...
(variant[:KEY=VAL[,KEY=VAL]][@library], keys of slx_set_tuning; @library runs the arm on another build of libslx.so loaded
into the same process, so that two builds are compared launch by launch on one box)
All arms run in one process on one device, round-robin, and the median over rounds is reported."""
import importlib, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
arms = []
apis = {"": api}
def api_for(path):
    if path not in apis:
        import importlib.util
        sp = importlib.util.spec_from_file_location("slx_api_%d" % len(apis), api.__file__)
        m = importlib.util.module_from_spec(sp); sp.loader.exec_module(m)
        m.LIB_PATH = os.path.join(ROOT, path)
        apis[path] = m
    return apis[path]
for a in sys.argv[1:]:
    spec_, _, libpath = a.partition("@")
    v, _, envs = spec_.partition(":")
    env = {k: int(v_) for k, v_ in (e.split("=") for e in envs.split(",") if e)}
    arms.append((a, int(v), env, libpath))
cfg = os.environ.get("AB_CONFIG", "C4")
n_sets = int(os.environ.get("AB_SETS", "32"))
spec = synth.make_spec(cfg)
H, W = spec["height"], spec["width"]
n_phase, n_gray = synth.n_planes(spec)
pitch = W + int(os.environ.get("AB_PAD", "0"))           # AB_PAD: extra bytes per image row (moves the planes' relative alignment)
phase = torch.randint(0, 256, (n_sets, n_phase, H, pitch), dtype=torch.uint8, device="cuda")[..., :W] if n_phase else None
gray = torch.randint(0, 256, (n_sets, n_gray, H, pitch), dtype=torch.uint8, device="cuda")[..., :W] if n_gray else None
z = torch.empty((n_sets, H, W), dtype=torch.float64, device="cuda")
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctxs = {}
for name, v, env, libpath in arms:
    c = api_for(libpath).Context(spec); c.set_variant(v); c.set_tuning(**env); ctxs[name] = c
def run(name, v, env, n):
    c = ctxs[name]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(n): c.decode_batch(n_sets, phase, gray, z, stream=s.cuda_stream)
    e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n
for name, v, env, _ in arms: run(name, v, env, 20)
res = {name: [] for name, _, _, _ in arms}
for r in range(int(os.environ.get("AB_ROUNDS", "7"))):
    for name, v, env, _ in arms: res[name].append(run(name, v, env, 30))
bytes_ = n_sets * H * W * synth.algorithmic_bytes_per_pixel(spec)
for name, _, _, _ in arms:
    m = statistics.median(res[name])
    print("%-40s median %7.1f us  min %7.1f  -> %5.2f TB/s (%.1f %% of 8 TB/s)" % (name, m, min(res[name]), bytes_ / m / 1e6, bytes_ / m / 1e6 / 8 * 100))
