#!/usr/bin/env python3
"""What binding reaches the fast kernel.  At the reference's own size (1280x1024, 6-bit Gray + 4-step):
  A  the decoder-class swap: CDecodeGray::Decode + CDecodePhase::Decode as TWO launches of the general kernel, each result
     (an f64 plane) copied to the host, merge + FillCoordinate left to the host loop -- slx::CDecodeGray / slx::CDecodePhase;
  B  slx::CCalculation::CalculateFirst / slx_decode in SLX_MODE_GRAY_PHASE: ONE fused launch (decode, merge, triangulation),
     depth copied to the host.
Frames resident in device memory in both; kernel time by slx_enable_timing, wall time per call including the copies."""
import importlib, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
ref = synth.make_spec("REF")
H, W = ref["height"], ref["width"]
ph, gr, _ = synth.render(ref, "sphere", seed=9, noise_sigma=1.0)
dph, dgr = torch.from_numpy(ph).cuda(), torch.from_numpy(gr).cuda()
torch.cuda.synchronize()
reps = 100


def run(spec, phase, gray, out, aux=()):
    with api.Context(spec, aux=aux) as ctx:
        ctx.set_frames(phase, gray)
        ctx.enable_timing(True)
        for _ in range(20):
            ctx.decode(); ctx.get_output(out)
        k, wall = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.decode()
            ctx.get_output(out)
            wall.append((time.perf_counter() - t0) * 1e6)
            k.append(ctx.last_decode_ms() * 1e3)
        return statistics.median(k), statistics.median(wall)


sg = dict(ref); sg["mode"] = synth.MODE_GRAY_ONLY
sp = dict(ref); sp["mode"] = synth.MODE_PHASE_ONLY
kg, wg = run(sg, None, dgr, "gray")
kp, wp = run(sp, dph, None, "pix")
kf, wf = run(ref, dph, dgr, "z")
print(json.dumps({"size": "%dx%d" % (W, H),
                  "A_decoder_classes": {"gray_kernel_us": round(kg, 1), "phase_kernel_us": round(kp, 1), "kernels_us": round(kg + kp, 1),
                                        "wall_us_with_two_f64_planes_to_host": round(wg + wp, 1)},
                  "B_fused": {"kernel_us": round(kf, 1), "wall_us_with_depth_to_host": round(wf, 1)},
                  "note": "A still leaves the merge (R/CCalculation.cpp:561-589) and FillCoordinate (:666-785) to the host CPU"}))
