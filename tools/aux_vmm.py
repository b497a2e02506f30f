#!/usr/bin/env python3
"""C4 x 16 with x, y, U, k beside z (the launch with two speeds, tools/aux_layout.py) on buffers from HIP's virtual-memory API: one
physical allocation per plane, one for all five planes, against torch's allocator -- does an allocation the process makes
explicitly (hipMemCreate: one physical object of exactly that size) always land in the fast mode?  Run several times."""
import ctypes, importlib, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
vmm = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "libvmm.so"))
vmm.vmm_alloc.restype = ctypes.c_void_p
vmm.vmm_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
vmm.vmm_free.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
spec = synth.make_spec("C4"); H, W = spec["height"], spec["width"]; n = 16
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ph = torch.randint(0, 256, (n, 12, H, W), dtype=torch.uint8, device="cuda")
plane = n * H * W * 8
NAMES = ("z", "x", "y", "U", "k")

def alloc(nbytes):
    g, m = ctypes.c_size_t(), ctypes.c_size_t()
    p = vmm.vmm_alloc(nbytes, ctypes.byref(g), ctypes.byref(m))
    assert p, "vmm_alloc failed"
    return p, m.value, g.value

def torch_separate():
    ts = {k: torch.empty((n, 2, H, W), dtype=torch.int32, device="cuda") if k == "k" else torch.empty((n, H, W), dtype=torch.float64, device="cuda") for k in NAMES}
    return {k: v.data_ptr() for k, v in ts.items()}, lambda: ts.clear()
def torch_arena():
    a = torch.empty((5 * plane + (4 << 20),), dtype=torch.uint8, device="cuda")
    base = a.data_ptr() + (-a.data_ptr()) % (2 << 20)
    return {k: base + i * plane for i, k in enumerate(NAMES)}, lambda: None if a is None else None
def vmm_separate():
    got = [alloc(plane) for _ in NAMES]
    return {k: g[0] for k, g in zip(NAMES, got)}, lambda: [vmm.vmm_free(p, m) for p, m, _ in got]
def vmm_arena():
    p, m, g = alloc(5 * plane)
    return {k: p + i * plane for i, k in enumerate(NAMES)}, lambda: vmm.vmm_free(p, m)

with api.Context(spec) as c:
    gran = alloc(1)[2]
    print("allocation granularity %d KiB" % (gran >> 10))
    for label, make in (("torch, separate tensors", torch_separate), ("torch, one arena", torch_arena), ("hipMemCreate per plane", vmm_separate),
                        ("hipMemCreate, one arena", vmm_arena), ("torch, one arena (again)", torch_arena), ("hipMemCreate, one arena (again)", vmm_arena)):
        o, release = make()
        torch.cuda.synchronize()
        for _ in range(60): c.decode_batch_ex(n, ph, None, stream=s.cuda_stream, **o)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(40): c.decode_batch_ex(n, ph, None, stream=s.cuda_stream, **o)
            e1.record(s); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000 / 40)
        print("%-34s median %7.1f us  min %7.1f   z at %x" % (label, statistics.median(ts), min(ts), o["z"]), flush=True)
        release()
        torch.cuda.empty_cache()
