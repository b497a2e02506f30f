import importlib, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
spec = synth.make_spec("C4"); H, W = spec["height"], spec["width"]; n = 32
phase = torch.randint(0, 256, (n, 12, H, W), dtype=torch.uint8, device="cuda")
z = torch.empty((n, H, W), dtype=torch.float64, device="cuda")
s = torch.cuda.Stream(); torch.cuda.synchronize()
ctx = api.Context(spec); ctx.set_variant(2)
def run(stream, k):
    t0 = time.perf_counter()
    for _ in range(k): ctx.decode_batch(n, phase, None, z, stream=stream)
    ctx.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6
run(None, 200); run(s.cuda_stream, 200)
for r in range(4):
    print("own stream %.2f us   caller stream (event per launch) %.2f us" % (run(None, 300), run(s.cuda_stream, 300)))
