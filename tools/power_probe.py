#!/usr/bin/env python3
"""What the card's power management does while a decode kernel runs back to back (GPU box).
Usage: tools/power_probe.py [CONFIG:SETS ...]     (default: C4:32 C2:80 C5:4 C3:16 REF:32 REFGRAY:32)
For each workload: launches of the batch decode back to back for AB_SECONDS (default 5) while a thread reads `rocm-smi`
(socket power, shader / memory clock, temperatures; sysfs reads, no GPU context) a few times per second.  Prints per workload
the launch time, the median / min / max of every sampled quantity and the idle readings before and after; raw samples go to
gpurun_out/power_probe.json.  Answers whether a kernel's clock is held down by the power cap."""
import importlib, json, os, re, statistics, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")

SMI = ["/opt/rocm/bin/rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showtemp", "--showperflevel", "--showmaxpower", "--json"]


def sample():
    try:
        out = subprocess.run(SMI, capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out[out.index("{"):])
    except Exception as e:                         # the tool's output is evidence, not control flow: keep what went wrong
        return {"error": repr(e)}
    card = d.get("card0", d)
    flat = {}
    for k, v in card.items():
        m = re.search(r"-?\d+(\.\d+)?", str(v))
        flat[k] = float(m.group(0)) if m else v
    return flat


class Sampler(threading.Thread):
    def __init__(self, period):
        super().__init__(daemon=True)
        self.period, self.samples, self.stop_flag = period, [], False

    def run(self):
        while not self.stop_flag:
            t = time.time()
            s = sample()
            s["t"] = t
            self.samples.append(s)
            time.sleep(max(0.0, self.period - (time.time() - t)))


def summarize(samples):
    keys = sorted({k for s in samples for k, v in s.items() if isinstance(v, float) and k != "t"})
    out = {}
    for k in keys:
        vals = [s[k] for s in samples if isinstance(s.get(k), float)]
        if vals:
            out[k] = {"median": statistics.median(vals), "min": min(vals), "max": max(vals), "n": len(vals)}
    return out


def workload(cfg, n_sets, seconds, tune):
    spec = synth.make_spec(cfg)
    H, W = spec["height"], spec["width"]
    n_phase, n_gray = synth.n_planes(spec)
    phase = torch.randint(0, 256, (n_sets, n_phase, H, W), dtype=torch.uint8, device="cuda") if n_phase else None
    gray = torch.randint(0, 256, (n_sets, n_gray, H, W), dtype=torch.uint8, device="cuda") if n_gray else None
    z = torch.empty((n_sets, H, W), dtype=torch.float64, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    ctx = api.Context(spec)
    ctx.set_variant(2)
    if tune:
        ctx.set_tuning(**tune)
    smp = Sampler(0.2)
    times = []
    t_end = time.time() + seconds
    started = False
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(200):
            ctx.decode_batch(n_sets, phase, gray, z, stream=s.cuda_stream)
        e1.record(s)
        if not started:                             # sample only while launches are queued
            smp.start()
            started = True
        e1.synchronize()
        times.append(e0.elapsed_time(e1) * 1000.0 / 200)
    smp.stop_flag = True
    smp.join()
    kern = ctx.last_kernel()
    del ctx
    return {"config": cfg, "sets": n_sets, "tune": tune, "kernel": kern, "launch_us_median": statistics.median(times), "launch_us_first": times[0],
            "launch_us_last": times[-1], "blocks": len(times), "smi": summarize(smp.samples), "raw": smp.samples}


def main():
    seconds = float(os.environ.get("AB_SECONDS", "5"))
    jobs = sys.argv[1:] or ["C4:32", "C2:80", "C5:4", "C3:16", "REF:32", "REFGRAY:32"]
    report = {"idle_before": sample(), "workloads": []}
    for j in jobs:
        parts = j.split(":")
        tune = {k: int(v) for k, v in (e.split("=") for e in parts[2].split(",") if e)} if len(parts) > 2 else {}
        r = workload(parts[0], int(parts[1]), seconds, tune)
        report["workloads"].append(r)
        smi = r["smi"]
        print("%-22s %-44s %7.1f us (first block %7.1f, last %7.1f)" % (j, r["kernel"][:44], r["launch_us_median"], r["launch_us_first"], r["launch_us_last"]))
        for k, v in smi.items():
            print("      %-48s median %9.1f  min %9.1f  max %9.1f  (%d samples)" % (k[:48], v["median"], v["min"], v["max"], v["n"]))
        sys.stdout.flush()
        time.sleep(1.0)
    report["idle_after"] = sample()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "power_probe.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("idle before:", {k: v for k, v in report["idle_before"].items() if isinstance(v, float)})
    print("idle after: ", {k: v for k, v in report["idle_after"].items() if isinstance(v, float)})


if __name__ == "__main__":
    main()
