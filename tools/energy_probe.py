#!/usr/bin/env python3
"""Joules per wave instruction on this card (GPU box): tools/energybench keeps one instruction class issuing on every SIMD while this
script reads the socket power and the shader clock (rocm-smi).  Usage: tools/energy_probe.py [SECONDS_PER_CLASS] [CLASS ...]
Prints, per class: wave-instructions per second (chip-wide), median power, median shader clock, and the energy per wave
instruction above the `s_nop` run (waves resident, nothing executing): (P - P_nop) / rate.  A class that drives the card into its
power cap runs at a lower clock AND voltage than one that does not -- the clock column says which.  Raw numbers go to
gpurun_out/energy_probe.json."""
import json, os, statistics, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import smi_sample          # noqa: E402  (one reader of rocm-smi for the bench line and the tools)

EXE = os.path.join(ROOT, "tools", "energybench")
if not os.path.exists(EXE):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "energybench.hip"), "-o", EXE])
args = sys.argv[1:]
seconds = float(args.pop(0)) if args and args[0].replace(".", "").isdigit() else 4.0
classes = args or subprocess.run([EXE], capture_output=True, text=True).stdout.split("classes:")[1].split()
rows = []
for cls in classes:
    proc = subprocess.Popen([EXE, cls, str(seconds)], stdout=subprocess.PIPE, text=True)
    t0 = time.time()
    samples = []
    while proc.poll() is None:
        s = smi_sample(0)
        if s and time.time() - t0 > seconds * 0.5 and proc.poll() is None:       # the half the program times
            samples.append(s)
        time.sleep(0.05)
    out = proc.stdout.read().strip().splitlines()
    try:
        res = json.loads(out[-1])
    except Exception:
        res = {"class": cls, "error": out[-1:] or "no output"}
    for key in ("socket_w", "sclk_mhz", "cap_w"):
        vals = [s[key] for s in samples if key in s]
        res[key] = statistics.median(vals) if vals else None
    res["smi_samples"] = len(samples)
    rows.append(res)
    print(json.dumps(res), flush=True)
    time.sleep(0.5)
base = next((r for r in rows if r.get("class") == "s_nop" and r.get("socket_w")), None)
print("\n%-16s %10s %8s %8s %12s %12s" % ("class", "Ginstr/s", "W", "MHz", "cycles/instr", "nJ/instr"))
for r in rows:
    if not r.get("wave_instr_per_s") or not r.get("socket_w"):
        continue
    rate = r["wave_instr_per_s"]
    cyc = r["sclk_mhz"] * 1e6 * 1024 / rate if r.get("sclk_mhz") else float("nan")
    r["cycles_per_instr"] = cyc
    r["nj_per_wave_instr_above_nop"] = (r["socket_w"] - base["socket_w"]) / rate * 1e9 if base else None
    print("%-16s %10.1f %8.0f %8.0f %12.2f %12s" % (r["class"], rate / 1e9, r["socket_w"], r.get("sclk_mhz") or 0, cyc,
                                                   "%.2f" % r["nj_per_wave_instr_above_nop"] if base else "-"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "energy_probe.json"), "w"), indent=1)
