import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
for cfg, n in (("C4", 3), ("C3", 2), ("REF", 2), ("C2", 2)):
    spec = synth.make_spec(cfg); H, W = spec["height"], spec["width"]
    npz, ng = synth.n_planes(spec)
    ph = torch.randint(0, 256, (n, npz, H, W), dtype=torch.uint8, device="cuda")
    gr = torch.randint(0, 256, (n, ng, H, W), dtype=torch.uint8, device="cuda") if ng else None
    ref = torch.empty((n, H, W), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()           # the fills run on torch's stream, the decode on the context's own: order them
    with api.Context(spec) as c:
        c.decode_batch(n, ph, gr, ref); c.synchronize()
        for wv in (4, 8, 16, 30, 32, 64):
            for rows in (0, 3, 16):
                z = torch.full((n, H, W), -7.0, dtype=torch.float64, device="cuda")
                torch.cuda.synchronize()
                c.set_tuning(weave=wv, strip_rows=rows)
                c.decode_batch(n, ph, gr, z); c.synchronize()
                ok = torch.equal(z.view(torch.int64), ref.view(torch.int64))
                print(cfg, "weave", wv, "rows", rows, "OK" if ok else "MISMATCH", flush=True)
                assert ok
print("weave ok")
