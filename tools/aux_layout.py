#!/usr/bin/env python3
"""C4 x 16 with x, y, U, k beside z: launch time against HOW the output tensors were allocated (GPU box).  The bench entry C4+xyUk is
bimodal from run to run (0.65 / 0.76 of the peak); tools/aux_alias.py found no dependence on the spacing of planes carved from one
arena.  This one times separately allocated tensors (what bench.py does), an arena, and separate tensors with other allocations
in between, prints every tensor's address, and is meant to be run several times in fresh processes."""
import importlib, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
spec = synth.make_spec("C4"); H, W = spec["height"], spec["width"]; n = 16
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ph = torch.randint(0, 256, (n, 12, H, W), dtype=torch.uint8, device="cuda")
plane = n * H * W * 8
SHAPES = (("z", torch.float64, (n, H, W)), ("x", torch.float64, (n, H, W)), ("y", torch.float64, (n, H, W)), ("U", torch.float64, (n, H, W)),
          ("k", torch.int32, (n, 2, H, W)))
def separate(gap=0):
    outs, keep = {}, []
    for name, dt, shape in SHAPES:
        outs[name] = torch.empty(shape, dtype=dt, device="cuda")
        if gap:
            keep.append(torch.empty((gap,), dtype=torch.uint8, device="cuda"))
    return outs, keep
def arena(pad):
    a = torch.empty((5 * (plane + pad) + (4 << 20),), dtype=torch.uint8, device="cuda")
    base, off, outs = (-a.data_ptr()) % (2 << 20), 0, {}
    for name, dt, shape in SHAPES:
        outs[name] = a[base + off: base + off + plane].view(dt).view(shape)
        off += plane + pad
    return outs, [a]
with api.Context(spec) as c:
    layouts = [("separate tensors", lambda: separate()), ("separate, 3 MiB strangers between", lambda: separate(3 << 20)),
               ("separate, 700 MiB strangers between", lambda: separate(700 << 20)), ("arena, pad 0", lambda: arena(0)), ("arena, pad 1 MiB + 4 KiB", lambda: arena((1 << 20) + 4096)),
               ("separate tensors again", lambda: separate())]
    for label, make in layouts:
        o, keep = make()
        torch.cuda.synchronize()
        for _ in range(60): c.decode_batch_ex(n, ph, None, stream=s.cuda_stream, **o)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(40): c.decode_batch_ex(n, ph, None, stream=s.cuda_stream, **o)
            e1.record(s); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000 / 40)
        print("%-40s median %7.1f us  min %7.1f   in %x  %s" % (label, statistics.median(ts), min(ts), ph.data_ptr(),
              " ".join("%s=%x" % (k, v.data_ptr()) for k, v in o.items())), flush=True)
        del o, keep
        torch.cuda.empty_cache()
    print(c.last_kernel())
