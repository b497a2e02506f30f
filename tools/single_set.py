#!/usr/bin/env python3
"""Small launches of the decode -- the call the reference actually makes (CCalculation::CalculateFirst decodes ONE frame-set,
R/CCalculation.cpp:171-206) and the 150-row tile of an 8-way row split -- timed warm, per rows-per-item choice.
Usage: tools/single_set.py [--work C4x1,REFx1,REFx1+xyU,X4x1+xyU,C4tile32,C3x1,C5x1] [--rows 0,1,2,3,4,6,8] [--launches 200]
rows = 0 is the library's own choice.  Every launch takes the NEXT of --rotate distinct frame-sets and depth maps (default: as many
as put the rotation beyond the 256 MiB Infinity Cache, so that a launch reads HBM and not what the launch before left in cache;
--rotate 1 re-decodes one cache-resident working set: "Infinity-Cache resident", not an HBM figure).  Two numbers per arm: `kernel_us` = event-to-event time around single launches (what a
kernel trace calls the duration, plus the events' own cost), `back_to_back_us` = N dependent launches / N."""
import argparse, importlib, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")
shard = importlib.import_module("structured-light-calculation_amd.shard")

ap = argparse.ArgumentParser()
ap.add_argument("--work", default="C4x1,REFx1,REFx1+xyU,X4x1+xyU,C4tile32")
ap.add_argument("--rows", default="0,1,2,3,4,6,8")
ap.add_argument("--launches", type=int, default=200)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--rotate", type=int, default=0, help="distinct frame-sets rotated through (0: enough to exceed 512 MB in + out)")
ap.add_argument("--extra", default="", help="more tuning keys for every arm, e.g. tail_pct=-1")
a = ap.parse_args()
extra = {k: int(v) for k, v in (e.split("=") for e in a.extra.split(",") if e)}


def workload(name):
    """spec, n_sets, aux planes"""
    aux = ()
    if name.endswith("+xyU"):
        name, aux = name[:-4], ("x", "y", "U")
    if name.endswith("+xyUk"):
        name, aux = name[:-5], ("x", "y", "U", "k")
    if name.startswith("X4x"):                      # the reference's own mode at 1920 x 1200
        spec = dict(synth.make_spec("REF"))
        spec["width"], spec["height"] = 1920, 1200
        spec["calib"] = synth.scaled_calibration(1920, 1200, spec["proj_width"])
        return spec, int(name[3:]), aux
    if "tile" in name:                              # rank 0's tile of an 8-way row split, n frame-sets
        cfg, n = name.split("tile")
        spec, _, _ = shard.row_tile_spec(synth.make_spec(cfg), 8, 0)
        return spec, int(n), aux
    cfg, n = name.split("x")
    return synth.make_spec(cfg), int(n), aux


for wname in a.work.split(","):
    spec, n_sets, aux = workload(wname)
    H, W = spec["height"], spec["width"]
    n_phase, n_gray = synth.n_planes(spec)
    bytes_ = n_sets * H * W * (n_phase + n_gray + 8 + sum(4 * (spec["n_freq"] - 1) if p == "k" else 8 for p in aux))
    rotate = a.rotate if a.rotate > 0 else max(1, -(-512_000_000 // bytes_))
    hold = n_sets * rotate
    phase = torch.randint(0, 256, (hold, n_phase, H, W), dtype=torch.uint8, device="cuda") if n_phase else None
    gray = torch.randint(0, 256, (hold, n_gray, H, W), dtype=torch.uint8, device="cuda") if n_gray else None
    outs = {"z": torch.empty((hold, H, W), dtype=torch.float64, device="cuda")}
    for p in aux:
        outs[p] = (torch.empty((hold, spec["n_freq"] - 1, H, W), dtype=torch.int32, device="cuda") if p == "k"
                   else torch.empty((hold, H, W), dtype=torch.float64, device="cuda"))
    torch.cuda.synchronize()
    # every arm gets its own context; the arms run round-robin (a box's clock drifts: interleaving keeps the arms comparable)
    arms = []
    for rows in (int(r) for r in a.rows.split(",")):
        ctx = api.Context(spec)
        ctx.set_variant(2)
        ctx.set_tuning(strip_rows=rows, **extra)
        arms.append((rows, ctx, torch.cuda.ExternalStream(ctx.stream_handle()), [], []))

    turn = [0]

    def launch(ctx):
        r = (turn[0] % rotate) * n_sets
        turn[0] += 1
        ctx.decode_batch_ex(n_sets, None if phase is None else phase[r:r + n_sets], None if gray is None else gray[r:r + n_sets],
                            **{k: v[r:r + n_sets] for k, v in outs.items()})
    for _, ctx, _, _, _ in arms:
        for _ in range(60):
            launch(ctx)
        ctx.synchronize()
    per_round = max(10, a.launches // a.rounds)
    for _ in range(a.rounds):
        for rows, ctx, st, single, b2b in arms:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.enable_timing(False)
            e0.record(st)
            for _ in range(per_round):
                launch(ctx)
            e1.record(st)
            ctx.synchronize()
            b2b.append(e0.elapsed_time(e1) * 1e3 / per_round)
            ctx.enable_timing(True)
            for _ in range(per_round):
                launch(ctx)
                single.append(ctx.last_decode_ms() * 1e3)
    for rows, ctx, st, single, b2b in arms:
        k, b = statistics.median(single), statistics.median(b2b)
        print(json.dumps({"work": wname, "rows": rows, "kernel_us": round(k, 2), "kernel_us_min": round(min(single), 2), "back_to_back_us": round(b, 2),
                          "bytes": bytes_, "distinct_sets_rotated": rotate, "frac_kernel": round(bytes_ / k / 8e6, 3), "frac_b2b": round(bytes_ / b / 8e6, 3)}), flush=True)
        ctx.close()
