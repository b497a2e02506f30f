"""Synthetic fringe-image source and the named workload configurations.

Stands in for the reference's offline "virtual sensor" (R/CSensorV.cpp:60-133,
which imread()s pre-captured BMPs that are not in the repository): renders the
camera images a DynaFrame rig would capture of a simple scene, by running the
triangulation of R/CCalculation.cpp:686 backwards (depth -> projector column)
and evaluating the pattern model written in R/CDecodePhase.cpp:59-62.

R/ = /root/reference/DynaFrame/DynaFrame/.
"""
import math

import numpy as np

MODE_PHASE_ONLY, MODE_GRAY_ONLY, MODE_GRAY_PHASE, MODE_MULTIFREQ, MODE_MULTIFREQ_GRAYMASK = range(5)

# R/Result.yml:1-28, transcribed as data (camera 640x512, projector 1280x800).
RESULT_YML = {
    "cam": [1.2138714552009253e+003, 0., 3.1950000000000000e+002,
            0., 1.2159945377703152e+003, 2.5550000000000000e+002,
            0., 0., 1.],
    "pro": [2.0288057545415668e+003, 0., 6.1958898841564314e+002,
            0., 2.0319614890033101e+003, 6.6520739361244557e+002,
            0., 0., 1.],
    "rot": [9.9143473372566937e-001, -1.2723342704854930e-002, 1.2998186532253575e-001,
            2.5847502916207063e-002, 9.9467300669012182e-001, -9.9787355687128362e-002,
            -1.2801982407153850e-001, 1.0229235705783506e-001, 9.8648223416959957e-001],
    "trans": [-3.1747826732013134e+000, -9.2770189525198721e-001, 3.9430125669975382e+000],
}


def standard_gray_lut(bits):
    """lut[gray] = bin for the reflected Gray code g = b ^ (b >> 1)
    (exactly the table of R/Patterns/vGrayCode.txt at bits == 6)."""
    n = 1 << bits
    lut = np.zeros(n, dtype=np.int16)
    for b in range(n):
        lut[b ^ (b >> 1)] = b
    return lut


def scaled_calibration(width, height, proj_width, trans_scale=10.0):
    """R/Result.yml re-targeted at a width x height camera and a proj_width-wide
    projector: focal lengths scale with width/640 (projector: proj_width/1280),
    the camera principal point moves to the image centre, T is taken x10 so that
    depths read as millimetres."""
    s = width / 640.0
    sp = proj_width / 1280.0
    cam = list(RESULT_YML["cam"])
    cam[0] *= s
    cam[4] *= s
    cam[2] = (width - 1) / 2.0
    cam[5] = (height - 1) / 2.0
    pro = list(RESULT_YML["pro"])
    for i in (0, 2, 4, 5):
        pro[i] *= sp
    return {"cam": cam, "pro": pro, "rot": list(RESULT_YML["rot"]),
            "trans": [t * trans_scale for t in RESULT_YML["trans"]]}


def make_spec(name):
    """Workload configurations of BASELINE.json (C1..C5) plus the reference's
    compiled-in case (REF: 1280x1024, 6-bit Gray + 4-step at T = 40)."""
    def base(w, h, pw, mode, periods, n_steps=4, gray_bits=0):
        spec = {
            "name": name, "width": w, "height": h, "row_offset": 0, "proj_width": pw, "mode": mode,
            "n_freq": len(periods), "n_steps": n_steps, "periods": list(periods),
            "gray_bits": gray_bits, "gray_stripe": (pw // (1 << gray_bits)) if gray_bits else 0,
            "gray_lut": standard_gray_lut(gray_bits) if gray_bits else None,
            "fov_min": 100.0, "fov_max": 1000.0,
            "calib": scaled_calibration(w, h, pw),
        }
        return spec

    if name == "C1":        # 640x480 single-frequency 4-step (unit frequency: absolute)
        return base(640, 480, 1280, MODE_MULTIFREQ, [1280])
    if name == "C1x4":      # 640x480 in the reference's own mode: 6-bit Gray + 4-step, T = 1280/(1<<5)
        return base(640, 480, 1280, MODE_GRAY_PHASE, [1280 // (1 << 5)], gray_bits=6)
    if name == "REF":       # R/StaticParameters.cpp:4-18
        s = base(1280, 1024, 1280, MODE_GRAY_PHASE, [1280 // (1 << 5)], gray_bits=6)
        return s
    if name == "REFPHASE":  # the reference's phase decoder alone (CDecodePhase, R/CCalculation.cpp:546-559): 4 planes in, pix out
        return base(1280, 1024, 1280, MODE_PHASE_ONLY, [1280 // (1 << 5)])
    if name == "REFGRAY":   # the reference's Gray decoder alone (CDecodeGray, R/CCalculation.cpp:536-545): 12 planes in, stripe edge out
        s = base(1280, 1024, 1280, MODE_GRAY_ONLY, [], gray_bits=6)
        s["n_freq"], s["n_steps"] = 0, 4
        return s
    if name == "C2":
        return base(1280, 720, 1280, MODE_MULTIFREQ, [1280, 160, 20])
    if name == "C3":
        return base(1920, 1200, 1920, MODE_MULTIFREQ_GRAYMASK, [1920, 240, 30], gray_bits=6)
    if name == "C4":
        return base(1920, 1200, 1920, MODE_MULTIFREQ, [1920, 240, 30])
    if name == "C5":
        return base(4096, 3000, 4096, MODE_MULTIFREQ, [4096, 512, 64, 8], n_steps=8)
    raise KeyError(name)


def n_planes(spec):
    mode = spec["mode"]
    n_phase = 0 if mode == MODE_GRAY_ONLY else spec["n_freq"] * spec["n_steps"]
    n_gray = 2 * spec["gray_bits"] if mode in (MODE_GRAY_ONLY, MODE_GRAY_PHASE, MODE_MULTIFREQ_GRAYMASK) else 0
    return n_phase, n_gray


def algorithmic_bytes_per_pixel(spec):
    """each u8 input read once + one f64 depth written."""
    n_phase, n_gray = n_planes(spec)
    return n_phase + n_gray + 8


def projection_scalars(calib):
    """P = Kp [R T] and the scalars of R/CCalculation.cpp:151-164 (numpy, for the
    forward model only -- never used as a parity reference)."""
    Kp = np.asarray(calib["pro"], dtype=np.float64).reshape(3, 3)
    R = np.asarray(calib["rot"], dtype=np.float64).reshape(3, 3)
    T = np.asarray(calib["trans"], dtype=np.float64).reshape(3, 1)
    P = Kp @ np.hstack([R, T])
    cam = calib["cam"]
    fu, fv, cx, cy = cam[0], cam[4], cam[2], cam[5]
    return P, fu, fv, cx, cy


def scene_depth(spec, scene="tilted", rows=None):
    """z(v,u) in mm for a few simple scenes."""
    H, W = spec["height"], spec["width"]
    r0 = spec.get("row_offset", 0)
    v = (np.arange(H, dtype=np.float64) + r0)[:, None] if rows is None else rows
    u = np.arange(W, dtype=np.float64)[None, :]
    if scene == "plane":
        return np.full((H, W), 500.0) + 0 * u + 0 * v
    if scene == "tilted":
        return 450.0 + 0.08 * (u - W / 2) * (640.0 / W) + 0.05 * (v - H / 2) * (640.0 / W)
    if scene == "sphere":
        z = np.full((H, W), 600.0) + 0 * u + 0 * v
        rr = ((u - W * 0.5) ** 2 + (v - H * 0.5) ** 2) / (0.3 * H) ** 2
        bump = np.sqrt(np.clip(1.0 - rr, 0.0, None)) * 120.0
        return z - bump
    raise KeyError(scene)


def projector_column(spec, z):
    """Inverse of R/CCalculation.cpp:686: U = (z cC + cA) / (z cD + cB)."""
    P, fu, fv, cx, cy = projection_scalars(spec["calib"])
    H, W = z.shape
    r0 = spec.get("row_offset", 0)
    v = (np.arange(H, dtype=np.float64) + r0)[:, None]
    u = np.arange(W, dtype=np.float64)[None, :]
    cA = fu * fv * P[0, 3]
    cB = fu * fv * P[2, 3]
    cC = (u - cx) * fv * P[0, 0] + (v - cy) * fu * P[0, 1] + fu * fv * P[0, 2]
    cD = (u - cx) * fv * P[2, 0] + (v - cy) * fu * P[2, 1] + fu * fv * P[2, 2]
    return (z * cC + cA) / (z * cD + cB)


def render(spec, scene="tilted", seed=0x5EED, noise_sigma=0.0, U=None):
    """Render the camera images of one frame-set.

    Returns (phase_planes uint8 [F*N,H,W] or None, gray_planes uint8 [2G,H,W] or None, U_true).
    Pattern model: I_k = (uint8)((sin(2 pi (U mod T)/T + 2 pi k/N) + 1) * 127)
    (R/CDecodePhase.cpp:59-62 comments); Gray pair b = bit b (LSB first) of
    g = bin ^ (bin >> 1), bin = (int)(U/S): pattern 220/20, inverse 20/220.
    Pixels whose U falls off the projector are black in every image.
    """
    H, W = spec["height"], spec["width"]
    if U is None:
        U = projector_column(spec, scene_depth(spec, scene))
    pw = spec["proj_width"]
    lit = (U >= 0) & (U < pw)
    rng = np.random.default_rng(seed)
    n_phase, n_gray = n_planes(spec)
    phase = None
    if n_phase:
        N = spec["n_steps"]
        phase = np.zeros((n_phase, H, W), dtype=np.uint8)
        for f, T in enumerate(spec["periods"]):
            ph = 2.0 * math.pi * np.fmod(U, T) / T
            for k in range(N):
                val = (np.sin(ph + 2.0 * math.pi * k / N) + 1.0) * 127.0
                if noise_sigma > 0:
                    val = val + rng.normal(0.0, noise_sigma, size=val.shape)
                img = np.clip(val, 0, 255).astype(np.uint8)
                img[~lit] = 0
                phase[f * N + k] = img
    gray = None
    if n_gray:
        G, S = spec["gray_bits"], spec["gray_stripe"]
        b = np.clip((U / S).astype(np.int64), 0, (1 << G) - 1)
        g = b ^ (b >> 1)
        gray = np.zeros((n_gray, H, W), dtype=np.uint8)
        for bit in range(G):
            on = ((g >> bit) & 1).astype(bool)
            pat = np.where(on, 220.0, 20.0)
            inv = np.where(on, 20.0, 220.0)
            if noise_sigma > 0:
                pat = pat + rng.normal(0.0, noise_sigma, size=pat.shape)
                inv = inv + rng.normal(0.0, noise_sigma, size=inv.shape)
            pat = np.clip(pat, 0, 255).astype(np.uint8)
            inv = np.clip(inv, 0, 255).astype(np.uint8)
            pat[~lit] = 0
            inv[~lit] = 0
            gray[2 * bit] = pat
            gray[2 * bit + 1] = inv
    return phase, gray, U


def random_planes(spec, seed):
    """Uniform random bytes in every plane: exercises every branch of the
    decoders without any scene structure (property tests)."""
    H, W = spec["height"], spec["width"]
    rng = np.random.default_rng(seed)
    n_phase, n_gray = n_planes(spec)
    phase = rng.integers(0, 256, size=(n_phase, H, W), dtype=np.uint8) if n_phase else None
    gray = rng.integers(0, 256, size=(n_gray, H, W), dtype=np.uint8) if n_gray else None
    return phase, gray
