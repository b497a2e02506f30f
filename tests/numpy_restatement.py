"""A SECOND, independent restatement of the path in numpy -- test infrastructure only.

oracle/slx_oracle.c (the checker of the GPU tests) and the HIP kernels were written by one author from one reading of the
reference.  This file is a separate reading, array-at-a-time instead of pixel-at-a-time, in another language and with no
code in common with either, so that a shared misreading has a chance to show up as a mismatch (tests/test_second_restatement.py
compares it with the oracle bit for bit).  It does NOT pin parity: it was not produced by the reference either, and
cvFastArctan is again restated from the published OpenCV 2.4.x algorithm.  "Parity unpinned" stands.

Written from (R/ = /root/reference/DynaFrame/DynaFrame/, read as text):
  R/CDecodePhase.cpp:59-75      wrapped phase, cast by cast
  R/CDecodeGray.cpp:120-125     table fill m_gray2bin[gray] = bin
  R/CDecodeGray.cpp:159-171     threshold: saturating u8 difference > 0
  R/CDecodeGray.cpp:181-200     bit-pack (1 << binIdx), table, times the stripe
  R/CCalculation.cpp:134-166    P = ProMat [R T], cA, cB, cC(v,u), cD(v,u)
  R/CCalculation.cpp:562-587    Gray / phase merge
  R/CCalculation.cpp:678-706    depth + FOV          R/CCalculation.cpp:760-767  x, y
The BUILD-DEFINED pieces (N-step sums, temporal unwrap, Gray mask) follow DESIGN.md section 3.

numpy rounds every float32 / float64 operation separately and never contracts a*b+c, which is the arithmetic the reference
was compiled to (MSVC /fp:precise on x64).
"""
import numpy as np

F32, F64 = np.float32, np.float64


def fast_atan2_deg(y, x):
    """cv::fastAtan2 of OpenCV 2.4.x (scalar branch), degrees in [0, 360), float32 in and out."""
    y = np.asarray(y, dtype=F32)
    x = np.asarray(x, dtype=F32)
    scale = F32(180.0 / np.pi)
    p1 = F32(0.9997878412794807) * scale
    p3 = F32(-0.3258083974640975) * scale
    p5 = F32(0.1555786518463281) * scale
    p7 = F32(-0.04432655554792128) * scale
    eps = F32(2.2204460492503131e-16)                                # (float)DBL_EPSILON
    ax, ay = np.abs(x), np.abs(y)
    swap = ~(ax >= ay)
    num = np.where(swap, ax, ay)
    den = np.where(swap, ay, ax) + eps
    with np.errstate(divide="ignore", invalid="ignore"):
        c = (num / den).astype(F32)
    c2 = c * c
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    a = np.where(swap, F32(90.0) - a, a)
    a = np.where(x < 0, F32(180.0) - a, a)
    a = np.where(y < 0, F32(360.0) - a, a)
    return a.astype(F32)


def pix_from_sin_cos(sin_value, cos_value, period):
    """R/CDecodePhase.cpp:67-75."""
    x = fast_atan2_deg(sin_value, cos_value)
    pix = ((x / F32(360)).astype(F64) * F64(period)).astype(F32)     # float / int -> float; * (double) -> double; stored in a float
    pix = (pix.astype(F64) + F64(0.5)).astype(F32)                   # pix += 0.5 (a double literal)
    over = pix > F32(period)                                         # float against int
    pix = np.where(over, pix - F32(period), pix).astype(F32)
    return pix.astype(F64)                                           # (double)pix


def wrapped_phase(planes, period):
    """4 phase-shifted uint8 images -> pix, float64 [H, W] (R/CDecodePhase.cpp:59-75)."""
    g = [p.astype(F32) for p in planes[:4]]
    sin_value = (g[0] - g[2]) / F32(2)
    cos_value = (g[1] - g[3]) / F32(2)
    return pix_from_sin_cos(sin_value, cos_value, period)


def wrapped_phase_nstep(planes, period):
    """x1 (BUILD-DEFINED): float32 weighted sums, k ascending, then the same tail."""
    n = len(planes)
    if n == 4:
        return wrapped_phase(planes, period)
    sy = np.zeros(planes[0].shape, dtype=F32)
    sx = np.zeros(planes[0].shape, dtype=F32)
    for k in range(n):
        ang = 2.0 * np.pi * k / n
        c, s = np.cos(ang), np.sin(ang)
        c = 0.0 if abs(c) < 1e-9 else c
        s = 0.0 if abs(s) < 1e-9 else s
        g = planes[k].astype(F32)
        sy = sy + g * F32(c)
        sx = sx + g * F32(s)
    scale = F32(2.0) / F32(n)
    return pix_from_sin_cos(sy * scale, sx * scale, period)


def gray_table(rows, size):
    """rows of "bin gray" as the code file lists them -> table[gray] = bin (R/CDecodeGray.cpp:120-125)."""
    t = np.zeros(size, dtype=np.int16)
    for b, g in rows:
        t[g] = b
    return t


def gray_decode(planes, table, stripe):
    """2G uint8 images (pattern b, inverse b) -> left edge of the stripe, float64 [H, W]."""
    G = len(planes) // 2
    code = np.zeros(planes[0].shape, dtype=np.int64)
    for b in range(G):
        diff = np.clip(planes[2 * b].astype(np.int32) - planes[2 * b + 1].astype(np.int32), 0, 255).astype(np.uint8)   # cv::Mat - : saturating
        white = np.where(diff > 0, 255, 0)
        code += np.where(white == 255, 1 << b, 0)
    binv = np.asarray(table, dtype=np.int16)[code]
    return binv.astype(F64) * F64(stripe), binv


def merge_gray_phase(gray_val, phase_val, stripe, period):
    """R/CCalculation.cpp:564-587."""
    even = (gray_val / stripe).astype(np.int64) % 2 == 0             # (int)(grayVal / vGrayPeriod) % 2 == 0
    T = F64(period)
    ph_even = np.where(phase_val > T * 0.75, phase_val - T, phase_val)
    ph_odd = np.where(phase_val < T * 0.25, phase_val + T, phase_val) - 0.5 * T
    return gray_val + np.where(even, ph_even, ph_odd)


def calibration(calib, width, height, row_offset=0):
    """P, cA, cB and the cC / cD tables (R/CCalculation.cpp:134-166)."""
    K = np.asarray(calib["cam"], dtype=F64).reshape(3, 3)
    Kp = np.asarray(calib["pro"], dtype=F64).reshape(3, 3)
    RT = np.hstack([np.asarray(calib["rot"], dtype=F64).reshape(3, 3), np.asarray(calib["trans"], dtype=F64).reshape(3, 1)])
    P = np.zeros((3, 4), dtype=F64)
    for r in range(3):
        for col in range(4):
            acc = F64(0.0)
            for k in range(3):                                       # the product of two small dense matrices, k ascending
                acc = acc + Kp[r, k] * RT[k, col]
            P[r, col] = acc
    fu, fv, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    cA = fu * fv * P[0, 3]
    cB = fu * fv * P[2, 3]
    u = np.arange(width, dtype=F64)[None, :]
    v = (np.arange(height, dtype=F64) + row_offset)[:, None]
    cC = (u - cx) * fv * P[0, 0] + (v - cy) * fu * P[0, 1] + fu * fv * P[0, 2]
    cD = (u - cx) * fv * P[2, 0] + (v - cy) * fu * P[2, 1] + fu * fv * P[2, 2]
    return dict(P=P, cA=cA, cB=cB, cC=cC, cD=cD, fu=fu, fv=fv, cx=cx, cy=cy, u=u, v=v)


def triangulate(U, cal, fov_min, fov_max, valid=None):
    """R/CCalculation.cpp:678-706 and :760-767.  U == 0 -> z = 0 (the reference leaves it unset; defined 0)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        z = -(cal["cA"] - cal["cB"] * U) / (cal["cC"] - cal["cD"] * U)
    z = np.where((z < fov_min) | (z > fov_max), 0.0, z)
    z = np.where(U == 0, 0.0, z)
    if valid is not None:
        z = np.where(valid, z, 0.0)
    uc = cal["u"] - cal["cx"]
    vc = cal["v"] - cal["cy"]
    with np.errstate(invalid="ignore"):
        x = z * uc / cal["fu"]
        y = z * vc / cal["fv"]
    return z, x, y


def unwrap(pix, periods):
    """x2 (BUILD-DEFINED): U_1 = pix_1; k_f = (int)floor((U_{f-1} - pix_f)/T_f + 0.5); U_f = pix_f + k_f T_f."""
    U = pix[0]
    ks = []
    for f in range(1, len(periods)):
        T = periods[f]
        k = np.floor((U - pix[f]) / F64(T) + 0.5).astype(np.int32)
        U = pix[f] + (k * np.int32(T)).astype(F64)
        ks.append(k)
    return U, ks


def gray_mask(U, gray_val, stripe):
    """x3 (BUILD-DEFINED): |U - (gray + S/2)| <= S, then a 3-tap horizontal AND (neighbours outside the image do not veto)."""
    S = F64(stripe)
    ok = np.abs(U - (gray_val + S * 0.5)) <= S
    left = np.ones_like(ok)
    right = np.ones_like(ok)
    left[:, 1:] = ok[:, :-1]
    right[:, :-1] = ok[:, 1:]
    return ok & left & right


def pipeline(spec, phase, gray):
    """Every output of the mode, as a dict of arrays (names as in the C ABI)."""
    mode, H, W = spec["mode"], spec["height"], spec["width"]
    out = {}
    N = spec.get("n_steps", 4)
    periods = spec.get("periods", [])
    if mode != 1:
        pix = [wrapped_phase_nstep([phase[f * N + k] for k in range(N)], periods[f]) for f in range(spec["n_freq"])]
        out["pix"] = np.stack(pix)
    if mode in (1, 2, 4):
        gv, _ = gray_decode(gray, spec["gray_lut"], spec["gray_stripe"])
        out["gray"] = gv
    if mode < 2:
        return out
    cal = calibration(spec["calib"], W, H, spec.get("row_offset", 0))
    valid = None
    if mode == 2:
        U = merge_gray_phase(out["gray"], out["pix"][0], spec["gray_stripe"], periods[0])
    else:
        U, ks = unwrap(out["pix"], periods)
        if ks:
            out["k"] = np.stack(ks)
        if mode == 4:
            valid = gray_mask(U, out["gray"], spec["gray_stripe"])
    out["U"] = U
    out["mask"] = np.ones((H, W), dtype=np.uint8) if valid is None else valid.astype(np.uint8)
    out["z"], out["x"], out["y"] = triangulate(U, cal, spec["fov_min"], spec["fov_max"], valid)
    return out
