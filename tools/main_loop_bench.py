#!/usr/bin/env python3
"""The reference's main() at the reference's own size, end to end, through the drop-in (GPU box): a DynaFrame data directory is
written (parameters.yml, Patterns/vGrayCode.txt, 12 + 4 static images and N dynamic ones of 1280x1024 as 8-bit BMP) and
tests/cpp/dynaframe_data_dir -- Init, CalculateFirst, Result, CalculateOther over every dynaCam<i>.bmp with a point-cloud text file
per frame (R/main.cpp:42-44, R/CCalculation.cpp:171-320) -- runs over it as a child process.  Prints the wall time of the program,
per dynamic frame, and the bytes of text written.  Usage: tools/main_loop_bench.py [N_DYNAMIC=100] [DIR=/tmp/slx_main_loop]"""
import importlib, json, os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dynaframe_files import write_bmp, write_calibration_yaml
synth = importlib.import_module("structured-light-calculation_amd.synth")

n_dyn = int(sys.argv[1]) if len(sys.argv) > 1 else 100
d = sys.argv[2] if len(sys.argv) > 2 else "/tmp/slx_main_loop"
exe = os.path.join(ROOT, "tests", "cpp", "dynaframe_data_dir")
spec = synth.make_spec("REF")                                   # 1280 x 1024, 6-bit Gray + 4-step: R/StaticParameters.cpp
H, W = spec["height"], spec["width"]
shutil.rmtree(d, ignore_errors=True)
for sub in ("g/iFrame", "g/cFrame", "Patterns", "PointCloud"):
    os.makedirs(os.path.join(d, sub))
t0 = time.time()
ph, gr, _ = synth.render(spec, "sphere", seed=5, noise_sigma=2.0)
cal = spec["calib"]
write_calibration_yaml(os.path.join(d, "parameters.yml"), cal["cam"], cal["pro"], cal["rot"], cal["trans"])
rows = json.load(open(os.path.join(ROOT, "tests", "golden", "vGrayCode_rows.json")))["rows"]
with open(os.path.join(d, "Patterns", "vGrayCode.txt"), "w") as f:
    f.write("\n".join("%d %d" % (b, g) for b, g in rows) + "\n")
for i in range(12):
    write_bmp(os.path.join(d, "g/iFrame/vGrayCam%d.bmp" % i), gr[i], bits=8)
for i in range(4):
    write_bmp(os.path.join(d, "g/iFrame/vPhaseCam%d.bmp" % i), ph[i], bits=8)
rng = np.random.default_rng(8)
u = np.arange(W)[None, :] + 0.03 * np.arange(H)[:, None]
for f in range(n_dyn + 1):                                     # dynaCam0 is the tracker's first image
    img = 128 + 100 * np.sign(np.sin(2 * np.pi * (u + 1.7 * f) / 14.0)) + rng.normal(0, 6, (H, W))
    write_bmp(os.path.join(d, "g/cFrame/dynaCam%d.bmp" % f), np.clip(img, 0, 255).astype(np.uint8), bits=8)
print("data directory written in %.1f s: %d images of %dx%d" % (time.time() - t0, 16 + n_dyn + 1, W, H), flush=True)
for rep in range(2):                                           # the second run has the files in the page cache and the GPU warm
    t1 = time.time()
    out = subprocess.check_output([exe, d, "g", "1280", str(spec["fov_min"]), str(spec["fov_max"])]).decode().strip()
    dt = time.time() - t1
    n_done = int(out.split("dynamic frames")[1]) if "dynamic frames" in out else n_dyn
    texts = [os.path.getsize(os.path.join(d, "PointCloud", n)) for n in os.listdir(os.path.join(d, "PointCloud"))]
    print("run %d: %s -- %.2f s wall for the program (process start, GPU context, %d image files read, 1 static + %d dynamic frames, "
          "%d point-cloud files = %.0f MB of text): %.1f ms per dynamic frame all in" % (rep, out, dt, 16 + n_dyn + 1, n_done, len(texts), sum(texts) / 1e6, dt / max(n_done, 1) * 1e3), flush=True)
shutil.rmtree(d, ignore_errors=True)
