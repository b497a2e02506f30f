#!/usr/bin/env python3
"""Differential fuzz of the batch decode against the oracle (GPU box): random tile shapes (ragged widths included), modes, frequency
/ step / Gray-bit counts, frame-set counts, row pitches, plane strides, optional planes, kernel variants and launch tunings, on
unstructured bytes.  Usage: tools/fuzz_parity.py [SECONDS] [SEED] [--cases N]     (default 90 s, seed from the clock; --cases: exactly N cases)
Prints one line per case class and a JSON line for every mismatch or unexpected error (the case can be replayed from its seed);
exit code 1 if anything differed.  The parity tests in tests/ pin chosen geometries; this walks the space between them."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import oracle as O                       # the checker
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")

PROFILE = os.environ.get("FUZZ_PROFILE", "wide")    # "strip": dword-aligned tiles, more rows and frame-sets, variants 0 / 2 -- the fast kernels' plans
_argv = sys.argv[1:]
CASES = None                             # --cases N: exactly N cases whatever the clock says (a run that two boxes repeat case for case)
if "--cases" in _argv:
    _k = _argv.index("--cases")
    CASES = int(_argv[_k + 1])
    del _argv[_k:_k + 2]
budget = float(_argv[0]) if _argv else 90.0
seed0 = int(_argv[1]) if len(_argv) > 1 else int(time.time())
print("fuzz_parity: %s, seed %d" % ("--cases %d" % CASES if CASES is not None else "%.0f s" % budget, seed0), flush=True)


def more(done):
    """Another case?  By count when --cases was given, else by the clock."""
    return done < CASES if CASES is not None else time.time() < t_end


def make_case(rng):
    mode = int(rng.choice([2, 3, 3, 3, 4, 4, 0, 1]))        # 0 / 1: the decoder objects alone (CDecodePhase::Decode -> pix, CDecodeGray::Decode -> gray)
    strip = PROFILE == "strip"
    W = int(rng.integers(1, 321 if strip else 161)) * 4 if (strip or rng.random() < 0.75) else int(rng.integers(4, 700))
    H = int(rng.integers(1, 420 if strip else 140))
    pw = int(rng.choice([W, 1280, 1920, 4096]))
    pw = max(pw, 64)
    if mode == 2:
        G = int(rng.integers(1, 9))
        periods = [max(2, pw // (1 << max(G - 1, 0)))]
        n_steps = 4
    elif mode == 1:
        G = int(rng.choice([6, 6, 1, 3, 8, 10]))
        periods = []
        n_steps = 4
    else:
        F = int(rng.integers(1, 5))
        periods = [min(pw, 1 << 14)]
        for _ in range(F - 1):
            periods.append(max(2, periods[-1] // int(rng.integers(2, 11))))
        n_steps = int(rng.choice([4, 4, 4, 4, 8, 8] if strip else [4, 4, 4, 4, 3, 5, 8, 8, 16]))
        G = int(rng.choice([6, 6, 6, 1, 3, 5, 8])) if mode == 4 else 0
        if mode == 0 and rng.random() < 0.6:
            periods, n_steps = periods[:1], 4                      # the reference's decoder: one frequency, four steps (the DMA-ring kernel)
    spec = {"name": "fuzz", "width": W, "height": H, "row_offset": int(rng.integers(0, 3000)) if rng.random() < 0.3 else 0, "proj_width": pw, "mode": mode,
            "n_freq": len(periods), "n_steps": n_steps, "periods": periods, "gray_bits": G,
            "gray_stripe": max(1, pw // (1 << G)) if G else 0, "gray_lut": synth.standard_gray_lut(G) if G else None,
            "fov_min": 100.0, "fov_max": 1000.0, "calib": synth.scaled_calibration(W, max(H, 2), pw)}
    case = {"spec": spec, "n_sets": int(rng.integers(1, 14 if strip else 7)), "pitch_pad": int(rng.choice([0, 0, 4, 12, 64, 100])),
            "plane_pad": int(rng.choice([0, 0, 0, 2, 16, 1000])), "variant": int(rng.choice([0, 2] if strip else [0, 0, 0, 2, 3, 1])),
            "tune": {}, "aux": []}
    if rng.random() < 0.6:
        for key, choices in (("strip_rows", [0, 1, 2, 3, 5, 8, 16, 20]), ("weave", [0, 1, 2, 4, 8]), ("stream", [0, 1, 2, 2]), ("stream_rows", [0, 1, 2, 3, 5, 16]),
                             ("tiers", [0, 1, 2, 3]), ("tail_pct", [0, 10, 30]), ("strip_waves", [0, 1, 2, 4]), ("gray_plain", [0, 0, 1]), ("plain_order", [0, 1])):
            if rng.random() < 0.4:
                case["tune"][key] = int(rng.choice(choices))
    if rng.random() < 0.5 and mode >= 2:
        pool = ["x", "y", "U"] + (["mask"] if mode in (3, 4) else []) + (["k"] if mode in (3, 4) and len(periods) > 1 else [])
        case["aux"] = [p for p in pool if rng.random() < 0.6]
    return case


def run_case(case, seed):
    spec, n = case["spec"], case["n_sets"]
    H, W = spec["height"], spec["width"]
    rng = np.random.default_rng(seed)
    n_phase, n_gray = synth.n_planes(spec)
    pitch = W + case["pitch_pad"]
    if pitch % 4 and W % 4 == 0:
        pitch += 4 - pitch % 4
    def planes(count):
        if not count:
            return None, None
        host = rng.integers(0, 256, size=(n, count, H, pitch), dtype=np.uint8)
        return host, torch.from_numpy(host).cuda()[..., :W]
    ph_h, ph = planes(n_phase)
    gr_h, gr = planes(n_gray)
    hw = H * W
    pstride = hw + case["plane_pad"] if case["plane_pad"] else 0
    if pstride % 2:
        pstride += 1
    per = pstride or hw
    F = spec["n_freq"]
    primary = {0: "pix", 1: "gray"}.get(spec["mode"], "z")     # what the batch call's first output holds
    outs, shapes = {}, {"z": (F if primary == "pix" else 1, torch.float64), "x": (1, torch.float64), "y": (1, torch.float64), "U": (1, torch.float64), "mask": (1, torch.uint8), "k": (max(F - 1, 1), torch.int32)}
    for name in ["z"] + case["aux"]:
        planes_per_set, dt = shapes[name]
        outs[name] = torch.full((n * planes_per_set * per + 64,), -7 if dt != torch.uint8 else 9, dtype=dt, device="cuda")
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        ctx.set_variant(case["variant"])
        if case["tune"]:
            ctx.set_tuning(**case["tune"])
        ctx.decode_batch_ex(n, ph, gr, plane_stride=pstride, row_stride=pitch, **outs)
        ctx.synchronize()
        kernel = ctx.last_kernel()
    torch.cuda.synchronize()
    bad = []
    for s in range(n):
        ref = O.pipeline(spec, None if ph_h is None else ph_h[s][..., :W], None if gr_h is None else gr_h[s][..., :W], want=tuple([primary] + case["aux"]))
        ref["z"] = ref[primary]
        for name in ["z"] + case["aux"]:
            pps = shapes[name][0]
            got = outs[name].cpu().numpy()
            stacked = name == "k" or (name == "z" and primary == "pix")
            for q in range((F - 1 if name == "k" else F) if stacked else 1):
                g = got[(s * pps + q) * per:(s * pps + q) * per + hw].reshape(H, W)
                r = ref[name][q] if stacked else ref[name]
                if not np.array_equal(g, r, equal_nan=True):
                    bad.append((s, name, q, int(np.sum(~((g == r) | ((g != g) & (r != r)))))))
    return kernel, bad


def run_big(seed):
    """FUZZ_PROFILE=big: launches that fill the chip many times over -- where the planner takes the stream kernel by itself -- on random
    widths / heights / frame-set counts of the Gray-free 4-step class: the automatic plan, the strip kernel (stream=1) and the stream
    kernel with another item length must agree bit for bit on every frame-set, and three frame-sets are compared with the oracle."""
    rng = np.random.default_rng(seed)
    W = int(rng.integers(64, 513)) * 4
    H = int(rng.integers(150, 1301))
    F = int(rng.integers(1, 5))
    n = int(max(2, min(48, rng.integers(20, 120) * 1000000 // (W * H))))
    periods = [min(W, 1 << 14)]
    for _ in range(F - 1):
        periods.append(max(2, periods[-1] // int(rng.integers(2, 11))))
    spec = {"name": "fuzz", "width": W, "height": H, "row_offset": int(rng.integers(0, 2000)) if rng.random() < 0.3 else 0, "proj_width": W, "mode": 3,
            "n_freq": F, "n_steps": 4, "periods": periods, "gray_bits": 0, "gray_stripe": 0, "gray_lut": None,
            "fov_min": 100.0, "fov_max": 1000.0, "calib": synth.scaled_calibration(W, H, W)}
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    ph = torch.randint(0, 256, (n, 4 * F, H, W), dtype=torch.uint8, device="cuda", generator=g)
    outs = []
    kernels = []
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        for tune in ({}, {"stream": 1}, {"stream": 2, "stream_rows": int(rng.choice([2, 3, 4, 7]))}):
            ctx.set_tuning(stream=0, stream_rows=0)
            if tune:
                ctx.set_tuning(**tune)
            z = torch.full((n, H, W), -7.0, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            for rep in range(2):                                     # twice: the queue counters carry over
                ctx.decode_batch(n, ph, None, z)
            ctx.synchronize()
            torch.cuda.synchronize()
            outs.append(z)
            kernels.append(ctx.last_kernel().split(":")[0])
    bad = []
    for a in (1, 2):
        if not torch.equal(outs[0].view(torch.int64), outs[a].view(torch.int64)):
            bad.append(("plan %d differs from the automatic plan" % a, kernels))
    for s_ in sorted({0, n - 1, int(rng.integers(0, n))}):
        ref = O.pipeline(spec, ph[s_].cpu().numpy(), None, want=("z",), threads=16)["z"]
        if not np.array_equal(outs[0][s_].cpu().numpy(), ref, equal_nan=True):
            bad.append(("frame-set %d differs from the oracle" % s_, kernels))
    return "big: " + " / ".join(kernels), bad, {"W": W, "H": H, "F": F, "n_sets": n, "periods": periods}


def run_refstream(seed):
    """FUZZ_PROFILE=refstream: the reference's own mode (6 Gray bits on the ring + one 4-step frequency, R/CCalculation.cpp:525-592) on
    random widths / heights / frame-set counts / row offsets / Gray tables: the automatic plan (slx_gstream_kernel from 8 items per
    resident wave on, else the strip kernel), the strip kernel (stream=1) and the stream kernel forced with a random item length must
    agree bit for bit on every frame-set -- twice in a row (the queue counters carry over) -- and three frame-sets are compared with the
    oracle."""
    rng = np.random.default_rng(seed)
    W = int(rng.integers(16, 513)) * 4
    if rng.random() < 0.6:
        W = max(64, W // 64 * 64)                                    # (an odd number of quads per row makes more chunk columns than the kernel has queues: strip kernel)
    H = int(rng.integers(20, 1301))
    n = int(max(2, min(40, rng.integers(4, 90) * 1000000 // (W * H))))
    pw = int(rng.choice([W, 1280, 1920]))
    std = rng.random() < 0.75
    lut = synth.standard_gray_lut(6) if std else rng.permutation(64).astype(np.int16)
    spec = {"name": "fuzz", "width": W, "height": H, "row_offset": int(rng.integers(0, 2000)) if rng.random() < 0.3 else 0, "proj_width": pw, "mode": 2,
            "n_freq": 1, "n_steps": 4, "periods": [max(2, pw // 32)], "gray_bits": 6, "gray_stripe": max(1, pw // 64), "gray_lut": lut,
            "fov_min": 100.0, "fov_max": 1000.0, "calib": synth.scaled_calibration(W, H, pw)}
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    ph = torch.randint(0, 256, (n, 4, H, W), dtype=torch.uint8, device="cuda", generator=g)
    gr = torch.randint(0, 256, (n, 12, H, W), dtype=torch.uint8, device="cuda", generator=g)
    if rng.random() < 0.5:                                           # clean stripes and exact ties in a part of the Gray planes
        gr[:, :, :, : W // 2] = torch.where(gr[:, :, :, : W // 2] > 127, 220, 20).to(torch.uint8)
        gr[:, 1::2, :, : W // 5] = gr[:, 0::2, :, : W // 5]
    outs, kernels = [], []
    rows = int(rng.choice([0, 1, 1, 2, 3, 5, 8, 16]))
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        for tune in ({}, {"stream": 1}, {"stream": 2, "stream_rows": rows}):
            ctx.set_tuning(stream=0, stream_rows=0)
            if tune:
                ctx.set_tuning(**tune)
            z = torch.full((n, H, W), -7.0, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            for rep in range(2):
                ctx.decode_batch(n, ph, gr, z)
            ctx.synchronize()
            torch.cuda.synchronize()
            outs.append(z)
            kernels.append(ctx.last_kernel().split(":")[0])
    bad = []
    # (a width whose quads per row share no factor with 64 makes more chunk columns than there are queues: the plan then keeps the strip kernel)
    for a in (1, 2):
        if not torch.equal(outs[0].view(torch.int64), outs[a].view(torch.int64)):
            bad.append(("plan %d differs from the automatic plan" % a, kernels))
    for s_ in sorted({0, n - 1, int(rng.integers(0, n))}):
        ref = O.pipeline(spec, ph[s_].cpu().numpy(), gr[s_].cpu().numpy(), want=("z",), threads=16)["z"]
        if not np.array_equal(outs[0][s_].cpu().numpy(), ref, equal_nan=True):
            bad.append(("frame-set %d differs from the oracle" % s_, kernels))
    return "refstream: " + " / ".join(kernels), bad, {"W": W, "H": H, "n_sets": n, "rows": rows, "std_lut": bool(std), "row_offset": spec["row_offset"]}


def run_calib(seed):
    """FUZZ_PROFILE=calib: the triangulation's exactness arguments under calibrations nobody would ship -- entries of the camera /
    projector matrices, the rotation and the translation scaled by powers of two up to 2^+-40, negated, zeroed; FOV windows that
    are negative, huge, a single value, or empty of points -- on small tiles, every kernel variant, z / x / y / U against the oracle
    (divisions by zero and by denormals, infinities and NaNs in the depth included: the oracle divides the IEEE way)."""
    rng = np.random.default_rng(seed)
    name = str(rng.choice(["C2", "C1x4", "C3", "C1"]))
    spec = dict(synth.make_spec(name))
    W = int(rng.integers(4, 60)) * 4
    H = int(rng.integers(2, 60))
    spec["width"], spec["height"] = W, H
    cal = synth.scaled_calibration(W, H, spec["proj_width"])
    span = int(rng.choice([3, 12, 40]))                              # mild, strong, absurd
    for key in ("cam", "pro", "rot", "trans"):
        vals = list(cal[key])
        for i in range(len(vals)):
            r = rng.random()
            if r < 0.08:
                vals[i] = 0.0
            elif r < 0.25:
                vals[i] = vals[i] * float(2.0 ** int(rng.integers(-span, span + 1)))
            elif r < 0.33:
                vals[i] = -vals[i]
            elif r < 0.36:
                vals[i] = float(rng.normal()) * float(2.0 ** int(rng.integers(-20, 21)))
        cal[key] = vals
    spec["calib"] = cal
    kind = rng.random()
    if kind < 0.25:
        lo = float(rng.normal()) * 10.0 ** int(rng.integers(-3, 8))
        spec["fov_min"], spec["fov_max"] = lo, lo + abs(float(rng.normal())) * 10.0 ** int(rng.integers(-3, 8))
    elif kind < 0.35:
        spec["fov_min"] = spec["fov_max"] = float(rng.normal()) * 100.0
    elif kind < 0.45:
        spec["fov_min"], spec["fov_max"] = -1e300, 1e300
    spec["row_offset"] = int(rng.integers(0, 5000)) if rng.random() < 0.3 else 0
    ph, gr = synth.random_planes(spec, seed)
    if rng.random() < 0.3:                                           # structured input: flat planes make U hit exact values (0 among them)
        ph = np.full_like(ph, int(rng.integers(0, 256)))
    wants = ("z", "x", "y", "U")
    ref = O.pipeline(spec, ph, gr, want=wants)
    bad = []
    label = "calib: variants"
    for variant in (0, 1, 3, 2):
        try:
            got = api.decode_frameset(spec, ph, gr, want=wants, variant=variant)
            label += " %d" % variant
        except api.SlxError as e:
            if variant == 0:
                label = "calib: refused (%s)" % str(e)[14:60]
                break
            continue                                                 # a variant may decline the operands; the automatic one may only refuse the config
        for n_ in wants:
            if not np.array_equal(got[n_], ref[n_], equal_nan=True):
                bad.append(("variant %d: %s differs in %d elements" % (variant, n_, int(np.sum(~((got[n_] == ref[n_]) | ((got[n_] != got[n_]) & (ref[n_] != ref[n_])))))), name))
    return label, bad, {"config": name, "W": W, "H": H, "fov": [spec["fov_min"], spec["fov_max"]], "calib": cal}


def run_bigstrip(seed):
    """FUZZ_PROFILE=bigstrip: the strip kernel's other classes at scale -- Gray + phase (the reference's mode), the Gray mask, 8 steps,
    with and without the optional planes -- on random widths / heights / frame-set counts: the automatic plan and two random launch
    geometries (rows per item, weave, tiers, waves per workgroup, Gray planes off the ring) must agree bit for bit on every plane of
    every frame-set, and three frame-sets are compared with the oracle."""
    rng = np.random.default_rng(seed)
    kind = str(rng.choice(["gray_phase", "mask", "eight", "four_aux"]))
    W = int(rng.integers(32, 385)) * 4
    H = int(rng.integers(60, 900))
    n = int(max(2, min(24, rng.integers(8, 60) * 1000000 // (W * H))))
    pw = int(rng.choice([W, 1280, 1920]))
    if kind == "gray_phase":
        mode, G, n_steps, periods = 2, 6, 4, [max(2, pw // 32)]
    else:
        F = int(rng.integers(1, 5))
        periods = [min(pw, 1 << 14)]
        for _ in range(F - 1):
            periods.append(max(2, periods[-1] // int(rng.integers(2, 11))))
        mode, G, n_steps = (4, int(rng.choice([6, 6, 4])), 4) if kind == "mask" else (3, 0, 8 if kind == "eight" else 4)
    spec = {"name": "fuzz", "width": W, "height": H, "row_offset": 0, "proj_width": pw, "mode": mode, "n_freq": len(periods), "n_steps": n_steps,
            "periods": periods, "gray_bits": G, "gray_stripe": max(1, pw // (1 << G)) if G else 0, "gray_lut": synth.standard_gray_lut(G) if G else None,
            "fov_min": 100.0, "fov_max": 1000.0, "calib": synth.scaled_calibration(W, H, pw)}
    n_phase, n_gray = synth.n_planes(spec)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    ph = torch.randint(0, 256, (n, n_phase, H, W), dtype=torch.uint8, device="cuda", generator=g)
    gr = torch.randint(0, 256, (n, n_gray, H, W), dtype=torch.uint8, device="cuda", generator=g) if n_gray else None
    aux = [p for p in ("x", "y", "U") if rng.random() < 0.5] if (kind == "four_aux" or rng.random() < 0.25) else []
    if aux and mode in (3, 4) and rng.random() < 0.5:
        aux.append("mask")
    dts = {"z": torch.float64, "x": torch.float64, "y": torch.float64, "U": torch.float64, "mask": torch.uint8}
    results, kernels = [], []
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        ctx.set_variant(2)
        for t in range(3):
            ctx.set_tuning(strip_rows=0, weave=0, tiers=0, strip_waves=0, tail_pct=0, gray_plain=0, stream=1)
            if t:
                ctx.set_tuning(strip_rows=int(rng.choice([1, 2, 3, 5, 8, 12, 16])), weave=int(rng.choice([0, 1, 2, 4, 8])), tiers=int(rng.integers(0, 4)),
                               strip_waves=int(rng.choice([0, 0, 1, 2, 4])), tail_pct=int(rng.choice([0, 10, 40])), gray_plain=int(rng.random() < 0.2))
            outs = {name: torch.full((n, H, W), 5 if name == "mask" else -7, dtype=dts[name], device="cuda") for name in ["z"] + aux}
            torch.cuda.synchronize()
            ctx.decode_batch_ex(n, ph, gr, **outs)
            ctx.synchronize()
            torch.cuda.synchronize()
            results.append(outs)
            kernels.append(ctx.last_kernel().split(":")[1].strip()[:28])
    bad = []
    for t in (1, 2):
        for name in ["z"] + aux:
            a, b = results[0][name], results[t][name]
            if not torch.equal(a.view(torch.int64) if a.dtype == torch.float64 else a, b.view(torch.int64) if b.dtype == torch.float64 else b):
                bad.append(("plan %d: %s differs from the automatic plan" % (t, name), kernels))
    for s_ in sorted({0, n - 1, int(rng.integers(0, n))}):
        ref = O.pipeline(spec, ph[s_].cpu().numpy(), None if gr is None else gr[s_].cpu().numpy(), want=tuple(["z"] + aux), threads=16)
        for name in ["z"] + aux:
            if not np.array_equal(results[0][name][s_].cpu().numpy(), ref[name], equal_nan=True):
                bad.append(("frame-set %d: %s differs from the oracle" % (s_, name), kernels))
    return "bigstrip: " + kind + (" +aux" if aux else ""), bad, {"kind": kind, "W": W, "H": H, "n_sets": n, "periods": periods, "aux": aux}


t_end = time.time() + budget
stats, failures, i = {}, 0, 0
while PROFILE == "calib" and more(i):
    seed = seed0 * 100003 + i
    i += 1
    try:
        label, bad, what = run_calib(seed)
        if bad:
            failures += 1
            print(json.dumps({"MISMATCH": str(bad[:4]), "seed": seed, "case": what}), flush=True)
    except Exception as e:
        failures += 1
        label = "error"
        print(json.dumps({"ERROR": "%s: %s" % (type(e).__name__, e), "seed": seed}), flush=True)
    stats[label] = stats.get(label, 0) + 1
    if failures >= 15:
        break
while PROFILE == "bigstrip" and more(i):
    seed = seed0 * 100003 + i
    i += 1
    try:
        label, bad, what = run_bigstrip(seed)
        if bad:
            failures += 1
            print(json.dumps({"MISMATCH": str(bad[:4]), "seed": seed, "case": what}), flush=True)
    except Exception as e:
        failures += 1
        label = "error"
        print(json.dumps({"ERROR": "%s: %s" % (type(e).__name__, e), "seed": seed}), flush=True)
    stats[label] = stats.get(label, 0) + 1
while PROFILE == "refstream" and more(i):
    seed = seed0 * 100003 + i
    i += 1
    try:
        label, bad, what = run_refstream(seed)
        if bad:
            failures += 1
            print(json.dumps({"MISMATCH": str(bad[:4]), "seed": seed, "case": what}), flush=True)
    except Exception as e:
        failures += 1
        label = "error"
        print(json.dumps({"ERROR": "%s: %s" % (type(e).__name__, e), "seed": seed}), flush=True)
    stats[label] = stats.get(label, 0) + 1
while PROFILE == "big" and more(i):
    seed = seed0 * 100003 + i
    i += 1
    try:
        label, bad, what = run_big(seed)
        if bad:
            failures += 1
            print(json.dumps({"MISMATCH": str(bad[:4]), "seed": seed, "case": what}), flush=True)
    except Exception as e:
        failures += 1
        label = "error"
        print(json.dumps({"ERROR": "%s: %s" % (type(e).__name__, e), "seed": seed}), flush=True)
    stats[label] = stats.get(label, 0) + 1
while more(i):
    seed = seed0 * 100003 + i
    i += 1
    case = make_case(np.random.default_rng(seed))
    label = None
    try:
        kernel, bad = run_case(case, seed)
        label = kernel.split(":")[0]
        if bad:
            failures += 1
            spec = {k: v for k, v in case["spec"].items() if k not in ("gray_lut", "calib")}
            print(json.dumps({"MISMATCH": bad[:6], "seed": seed, "kernel": kernel, "case": dict(case, spec=spec)}), flush=True)
    except api.SlxError as e:
        # a configuration the library refuses is fine when it says so (variant 2 on an ineligible tile, a stride it cannot take)
        label = "refused: " + str(e)[:60]
    except Exception as e:
        failures += 1
        spec = {k: v for k, v in case["spec"].items() if k not in ("gray_lut", "calib")}
        print(json.dumps({"ERROR": "%s: %s" % (type(e).__name__, e), "seed": seed, "case": dict(case, spec=spec)}), flush=True)
        label = "error"
    stats[label] = stats.get(label, 0) + 1
print("fuzz_parity: %d cases, %d failures" % (i, failures))
for k, v in sorted(stats.items(), key=lambda kv: -kv[1]):
    print("  %6d  %s" % (v, k))
sys.exit(1 if failures else 0)
