#!/usr/bin/env python3
"""Benchmark of the depth-reconstruction hot path on MI355X.

Metric (BASELINE.json): depth frames/s and achieved HBM GB/s at 1920x1200, 3-frequency x
4-step.  One "step" = one pass of the fused decode over this rank's batch of frame-sets
(32 per GPU: configuration 4's 256 frame-sets over 8 GPUs), inputs already resident in HBM.
Ranks shard the batch by frame-set with no collective in the data path ("weak" scaling);
the RCCL depth-map gather of north_star is timed separately and reported as `with_gather`.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "structured-light-calculation_amd"
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_batch(torch, synth, spec, n_sets, device, seed):
    """Synthetic camera images of n_sets scenes, rendered on the GPU (same forward model as
    synth.render): phase uint8 [n_sets, F*N, H, W] and, for modes with a Gray code, gray uint8 [n_sets, 2G, H, W]."""
    import numpy as np
    H, W = spec["height"], spec["width"]
    N, periods = spec["n_steps"], spec["periods"]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n_sets, len(periods) * N, H, W), dtype=torch.uint8, device=device)
    n_gray = synth.n_planes(spec)[1]
    gray = torch.empty((n_sets, n_gray, H, W), dtype=torch.uint8, device=device) if n_gray else None
    scenes = ("tilted", "sphere", "plane")
    for s in range(n_sets):
        z = synth.scene_depth(spec, scenes[s % len(scenes)]) + 3.0 * (s // len(scenes))
        U = torch.from_numpy(np.ascontiguousarray(synth.projector_column(spec, z))).to(device)
        lit = (U >= 0) & (U < spec["proj_width"])
        for f, T in enumerate(periods):
            ph = 2.0 * math.pi * torch.fmod(U, float(T)) / float(T)
            for k in range(N):
                val = (torch.sin(ph + 2.0 * math.pi * k / N) + 1.0) * 127.0
                val = val + torch.randn(val.shape, generator=g, device=device, dtype=torch.float64) * 2.0
                img = val.clamp_(0, 255).to(torch.uint8)
                img[~lit] = 0
                out[s, f * N + k] = img
        if n_gray:
            G, S = spec["gray_bits"], spec["gray_stripe"]
            b = (U / S).to(torch.int64).clamp_(0, (1 << G) - 1)
            code = b ^ (b >> 1)
            for bit in range(G):
                on = ((code >> bit) & 1).bool()
                for inv in (0, 1):
                    val = torch.where(on ^ bool(inv), 220.0, 20.0) + torch.randn(U.shape, generator=g, device=device, dtype=torch.float64) * 2.0
                    img = val.clamp_(0, 255).to(torch.uint8)
                    img[~lit] = 0
                    gray[s, 2 * bit + inv] = img
    return out, gray


def cpu_baseline(spec, phase_np, gray_np=None, budget_s=12.0):
    """The oracle (CPU restatement, reference loop order, one thread) timed on whole frame-sets of
    the same workload until ~budget_s of CPU work; then once more on all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O   # checker / CPU baseline only
    O.build()
    n = 0
    t0 = time.perf_counter()
    while True:
        O.pipeline(spec, phase_np[n % len(phase_np)], None if gray_np is None else gray_np[n % len(phase_np)], want=("z",), threads=1, faithful_order=1)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 200:
            break
    single = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
              "sample": "%d frame-sets of %dx%d (%s), oracle/slx_oracle.c single thread, reference loop order, %.1f s" % (
                  n, spec["width"], spec["height"], spec["name"], dt)}
    cores = min(len(os.sched_getaffinity(0)), 16)      # a 1-GPU box gets a 16-core share of the host
    m = 0
    t0 = time.perf_counter()
    while True:
        O.pipeline(spec, phase_np[m % len(phase_np)], None if gray_np is None else gray_np[m % len(phase_np)], want=("z",), threads=cores, faithful_order=0)
        m += 1
        dt2 = time.perf_counter() - t0
        if dt2 >= budget_s / 3 or m >= 400:
            break
    multi = {"value": m / dt2, "unit": "frames/s", "cores": cores, "kind": "port",
             "sample": "%d frame-sets, row-parallel OpenMP, %.1f s" % (m, dt2)}
    return single, multi, O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # the chip needs ~100 launches (~35 ms) of this kernel after an idle spell before its clock settles
    # (tools/ramp.py: 504, 359, 331, 325, 317, 316 ... us per launch in blocks of 25), hence the warm-up default
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--sets-per-gpu", type=int, default=32)
    ap.add_argument("--config", default="C4")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true",
                    help="N > 1: skip the second timed region (decode + RCCL gather of the depth maps to rank 0, reported as with_gather)")
    ap.add_argument("--shard", choices=("framesets", "rows"), default="framesets",
                    help="how ranks split the batch: whole frame-sets (default), or a row tile of every frame-set (north_star's wording); "
                         "the per-GPU bytes are the same")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) on GPUs; gloo only to rehearse the multi-rank path on one GPU")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the decode has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    synth = importlib.import_module(PKG + ".synth")
    api = importlib.import_module(PKG + ".api")
    shard = importlib.import_module(PKG + ".shard")
    if not os.path.exists(api.LIB_PATH) and int(os.environ.get("LOCAL_RANK", "0")) == 0:
        import __graft_entry__      # an unbuilt tree (fresh checkout): compile the HIP library once, in-tree
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    api.lib()   # raises if the HIP library is missing: there is no other implementation

    spec = synth.make_spec(args.config)
    full_h = spec["height"]
    n_sets = args.sets_per_gpu
    if args.shard == "rows" and world > 1:
        # every rank decodes its row tile of all world * sets_per_gpu frame-sets (row_offset keeps v - cy right)
        spec, _, _ = shard.row_tile_spec(spec, world, rank)
        n_sets = args.sets_per_gpu * world
    H, W = spec["height"], spec["width"]
    n_phase, n_gray = synth.n_planes(spec)
    bytes_per_set = H * W * synth.algorithmic_bytes_per_pixel(spec)
    bytes_per_launch = n_sets * bytes_per_set

    t0 = time.perf_counter()
    phase, gray = make_batch(torch, synth, spec, n_sets, device, seed=0x5EED + 4 + rank)
    z = torch.empty((n_sets, H, W), dtype=torch.float64, device=device)
    torch.cuda.synchronize()
    if rank == 0:
        log("[bench] rendered %d frame-sets (%.2f GB in, %.2f GB out per step) in %.1f s" %
            (n_sets, phase.numel() / 1e9, z.numel() * 8 / 1e9, time.perf_counter() - t0))

    ctx = api.Context(spec, device=dev_index)
    ctx.set_variant(args.variant)
    stream = torch.cuda.Stream(device=device)      # an explicit stream: the library treats NULL as "my own stream"
    torch.cuda.set_stream(stream)
    sh = stream.cuda_stream
    assert sh != 0

    def step():
        ctx.decode_batch(n_sets, phase, gray, z, stream=sh)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The chip needs ~100 launches of this kernel after an idle spell before its clock settles (tools/ramp.py).  When the
    # caller asks for fewer warm-up steps than that, the difference runs here, untimed and reported as "settle_launches",
    # so that a short --warmup still measures the settled clock.
    settle = max(0, 100 - args.warmup)
    for _ in range(settle):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)                    # HIP events on the stream the kernel is launched on
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    fence()
    kernel_ms = ev0.elapsed_time(ev1) / args.steps          # average launch duration over the timed region
    coll_dev = device if args.backend == "nccl" else torch.device("cpu")
    t = torch.tensor([t_local, kernel_ms], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_max, kernel_ms_max = float(t[0]), float(t[1])

    # parity of what was just timed: frame-set 0 of rank 0 against the oracle
    result = {}
    gather = None
    if world > 1 and not args.no_gather:
        def gather():
            if args.backend == "nccl":
                return shard.gather_depth(z, dst=0)
            torch.cuda.synchronize()
            return shard.gather_depth(z.cpu(), dst=0)            # rehearsal only
        gather_error = None
        try:
            full = gather()
            fence()
            tg = time.perf_counter()
            reps = max(3, min(20, args.steps // 15))
            for _ in range(reps):
                step()
                full = gather()
            torch.cuda.synchronize()
            tg_local = time.perf_counter() - tg
            fence()
            tt = torch.tensor([tg_local], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        except Exception as e:      # the decode-only line above must still be reported
            full, reps, tt = None, 0, None
            gather_error = "%s: %s" % (type(e).__name__, e)
        gather = {"error": gather_error} if tt is None else {"value": world * args.sets_per_gpu * reps / float(tt[0]), "unit": "frames/s", "steps": reps,
                  "collective": "torch.distributed.gather (%s) of f64 depth maps to rank 0" % ("RCCL" if args.backend == "nccl" else args.backend),
                  "gathered_bytes_per_step": int(world * n_sets * H * W * 8),
                  "gathered_shape": list(full.shape) if (rank == 0 and full is not None) else None}

    if rank == 0:
        cpu_single = cpu_multi = None
        parity = None
        if not args.no_cpu_baseline and world == 1:
            sample = phase[: min(4, n_sets)].cpu().numpy()
            gsample = gray[: min(4, n_sets)].cpu().numpy() if gray is not None else None
            cpu_single, cpu_multi, O = cpu_baseline(spec, sample, gsample)
            ref = O.pipeline(spec, sample[0], None if gsample is None else gsample[0], want=("z",), threads=min(len(os.sched_getaffinity(0)), 16))["z"]
            got = z[0].cpu().numpy()
            parity = bool(np.array_equal(got, ref, equal_nan=True))
            if not parity:
                raise SystemExit("bench: frame-set 0 differs from the oracle -- refusing to report a number")
        achieved = bytes_per_launch / (kernel_ms_max * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                ent = json.load(open(tp)).get(args.config, {})
                traffic = ent["hbm_bytes_per_launch"] * n_sets / ent["sets_per_launch"]     # measured per frame-set, scaled to this launch
            except Exception:
                traffic = None
        result = {
            "metric": "depth_frames_per_sec", "value": world * args.sets_per_gpu * args.steps / t_max, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+f64", "data": "synthetic",
            "config": {"workload": "%s: %dx%d, %d-frequency x %d-step temporal unwrap%s + triangulation, %d frame-sets per GPU per step"
                                   % (args.config, W, H, spec["n_freq"], spec["n_steps"], " + %d-bit Gray mask" % spec["gray_bits"] if n_gray else "", n_sets),
                       "periods": spec["periods"], "sharding": ("by frame-set" if args.shard == "framesets" or world == 1 else "by row tile (%d rows of %d per GPU)" % (H, full_h)) + ", no data-path collective",
                       "kernel_variant": args.variant, "settle_launches": settle},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": ("slx_strip_kernel" if (args.variant in (0, 2) and ((spec["mode"] == 3 and spec["n_steps"] in (4, 8)) or (spec["mode"] in (2, 4) and spec["n_steps"] == 4))) else "slx_fused_kernel") + "<mode %d, F=%d, N=%d>" % (spec["mode"], spec["n_freq"], spec["n_steps"]), "launch_ms": kernel_ms_max,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
            "achieved_hbm_gbps_per_gpu": achieved,
            "cpu_baseline": cpu_single, "cpu_baseline_all_cores": cpu_multi,
            "parity_vs_oracle": parity, "with_gather": gather,
        }
        print(json.dumps(result), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
