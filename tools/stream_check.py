"""Stream kernel (slx_set_tuning stream=2): every rows-per-item choice gives the bits of the strip kernel's default plan, launch after
launch (the queue counters carry over between launches of one geometry and are zeroed when it changes).  GPU box."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
bad = 0
for cfg, shape, n in (("C4", None, 8), ("C2", None, 12), ("C1", None, 6), ("C4", (516, 71), 9), ("C4", (1920, 150), 16), ("C2", (64, 7), 40), ("C4", (1920, 1200), 1)):
    spec = dict(synth.make_spec(cfg))
    if shape:
        spec["width"], spec["height"] = shape
        spec["calib"] = synth.scaled_calibration(shape[0], shape[1], spec["proj_width"])
    H, W = spec["height"], spec["width"]
    npz, _ = synth.n_planes(spec)
    ph = torch.randint(0, 256, (n, npz, H, W), dtype=torch.uint8, device="cuda")
    ref = torch.full((n, H, W), -1.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with api.Context(spec) as c:
        c.set_variant(2)
        c.set_tuning(stream=1)
        c.decode_batch(n, ph, None, ref); c.synchronize()
        for rows in (0, 2, 3, 4, 7, 16):
            for rep in range(3):                       # the counters carry over from launch to launch
                z = torch.full((n, H, W), -7.0, dtype=torch.float64, device="cuda")
                torch.cuda.synchronize()
                c.set_tuning(stream=2, stream_rows=rows)
                c.decode_batch(n, ph, None, z); c.synchronize()
                ok = torch.equal(z.view(torch.int64), ref.view(torch.int64))
                if not ok:
                    bad += 1
                    d = (z.view(torch.int64) != ref.view(torch.int64))
                    print(cfg, shape, n, "rows", rows, "rep", rep, "MISMATCH", int(d.sum()), "unwritten", int((z == -7.0).sum()), "first", d.nonzero()[:3].tolist(), flush=True)
                    break
            else:
                print(cfg, shape, n, "rows", rows, "OK", flush=True)
print("stream ok" if not bad else "stream FAILED: %d" % bad)
sys.exit(1 if bad else 0)
