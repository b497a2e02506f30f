"""Generates tests/golden/*.npz and wrapped_phase_tables.json from the CPU oracle.

These are ORACLE-GENERATED regression vectors: the reference has no tests or golden
outputs and cannot be built in this image (OpenCV 2.4.9 absent), so they are NOT
reference-pinned ("parity unpinned").  They pin the oracle against accidental change and
let the GPU box check the HIP path against bytes that were produced here.
The two reference-held data files are transcribed separately (vGrayCode_rows.json,
result_yml.json).

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

synth = importlib.import_module("structured-light-calculation_amd.synth")
HERE = os.path.dirname(os.path.abspath(__file__))

PERIODS = [40, 20, 30, 160, 240, 1280, 1920, 4096, 8, 64, 512]


def exhaustive_planes():
    """511x511 image whose pixel (i,j) has I0-I2 = i-255 and I1-I3 = j-255."""
    d = np.arange(-255, 256)
    d02, d13 = np.meshgrid(d, d, indexing="ij")
    p = np.zeros((4, 511, 511), dtype=np.uint8)
    p[0] = np.maximum(d02, 0)
    p[2] = np.maximum(-d02, 0)
    p[1] = np.maximum(d13, 0)
    p[3] = np.maximum(-d13, 0)
    return p


def small_spec(name, w=64, h=48):
    spec = synth.make_spec(name)
    spec = dict(spec)
    spec["width"], spec["height"] = w, h
    spec["calib"] = synth.scaled_calibration(w, h, spec["proj_width"])
    return spec


WANT = {
    synth.MODE_GRAY_PHASE: ("z", "x", "y", "U", "pix", "gray"),
    synth.MODE_MULTIFREQ: ("z", "x", "y", "U", "pix", "k"),
    synth.MODE_MULTIFREQ_GRAYMASK: ("z", "x", "y", "U", "pix", "gray", "k", "mask"),
}


def main():
    planes = exhaustive_planes()
    tables = {}
    for T in PERIODS:
        spec = {"width": 511, "height": 511, "mode": synth.MODE_PHASE_ONLY, "n_freq": 1, "n_steps": 4, "periods": [T]}
        pix = O.pipeline(spec, planes, None, want=("pix",))["pix"][0]
        rng = np.random.default_rng(T)
        idx = rng.integers(0, 511, size=(256, 2))
        tables[str(T)] = {
            "sha256": hashlib.sha256(np.ascontiguousarray(pix).tobytes()).hexdigest(),
            "samples": [[int(i), int(j), float(pix[i, j])] for i, j in idx],
        }
    json.dump({"note": "oracle-generated; pix[i][j] for I0-I2 = i-255, I1-I3 = j-255; sha256 over the f64 511x511 table",
               "tables": tables}, open(os.path.join(HERE, "wrapped_phase_tables.json"), "w"), indent=0)

    for name, scene, noise in (("C1x4", "sphere", 2.0), ("C2", "tilted", 2.0), ("C3", "sphere", 2.0), ("C5", "tilted", 1.0)):
        spec = small_spec(name)
        ph, gr, _ = synth.render(spec, scene, seed=0x5EED + len(name), noise_sigma=noise)
        res = O.pipeline(spec, ph, gr, want=WANT[spec["mode"]])
        out = {"out_" + k: v for k, v in res.items()}
        if ph is not None:
            out["phase"] = ph
        if gr is not None:
            out["gray_planes"] = gr
        np.savez_compressed(os.path.join(HERE, "scene_%s.npz" % name), **out)
        print(name, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
