#!/usr/bin/env python3
"""State-machine fuzz of one context through the C ABI (GPU box): random SEQUENCES of the calls a host loop can make -- frames set
one at a time from host / strided host / device memory, decodes on the context's or a caller's stream, batch decodes into the
caller's buffers in between, every output plane read back, point clouds, the tracker begun and stepped in all its feeding modes,
variants and tunings switched under way, calls made too early or with bad arguments -- with a model of what the context must hold
after each call, built from the oracle.  A call the model says is valid must succeed and leave the oracle's bits; a call the
model says is invalid must return an error code (and change nothing).  Usage: tools/fuzz_api.py [SECONDS] [SEED] [--cases N]."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import oracle as O                       # the checker
synth = importlib.import_module("structured-light-calculation_amd.synth")
api = importlib.import_module("structured-light-calculation_amd.api")

_argv = sys.argv[1:]
CASES = None                             # --cases N: exactly N cases whatever the clock says (a run that two boxes repeat case for case)
if "--cases" in _argv:
    _k = _argv.index("--cases")
    CASES = int(_argv[_k + 1])
    del _argv[_k:_k + 2]
budget = float(_argv[0]) if _argv else 90.0
seed0 = int(_argv[1]) if len(_argv) > 1 else int(time.time())
print("fuzz_api: %s, seed %d" % ("--cases %d" % CASES if CASES is not None else "%.0f s" % budget, seed0), flush=True)


def more(done):
    """Another case?  By count when --cases was given, else by the clock."""
    return done < CASES if CASES is not None else time.time() < t_end


class Mismatch(Exception):
    pass


def one_case(seed):
    rng = np.random.default_rng(seed)
    name = str(rng.choice(["C1x4", "C2", "C3", "C1"]))
    spec = dict(synth.make_spec(name))
    w = int(rng.integers(6, 120)) * 4 if rng.random() < 0.8 else int(rng.integers(23, 300))
    h = int(rng.integers(3, 110))
    spec["width"], spec["height"] = w, h
    spec["calib"] = synth.scaled_calibration(w, h, spec["proj_width"])
    mode, F = spec["mode"], spec["n_freq"]
    n_phase, n_gray = synth.n_planes(spec)
    planes = ["z"] + [p for p in ("U", "x", "y") if rng.random() < 0.8] + (["mask"] if mode in (3, 4) and rng.random() < 0.5 else []) + \
             (["k"] if mode in (3, 4) and F > 1 and rng.random() < 0.5 else []) + (["pix"] if rng.random() < 0.3 else []) + \
             (["gray"] if spec["gray_bits"] and rng.random() < 0.3 else [])
    aux = [p for p in planes if p != "z"]
    log = []                                       # the calls made, for the report

    def say(*a):
        log.append(" ".join(str(x) for x in a))
        if len(log) > 60:
            del log[0]

    def expect_error(fn, what):
        try:
            fn()
        except api.SlxError:
            return
        raise Mismatch("%s was accepted" % what)

    def same(tag, got, want):
        if not np.array_equal(got, want, equal_nan=True):
            raise Mismatch("%s differs (%d elements)" % (tag, int(np.sum(~((got == want) | ((got != got) & (want != want)))))))

    frames = {"p": [None] * n_phase, "g": [None] * n_gray}          # what the context holds, as host copies
    keep = []                                                        # device tensors the context borrows
    cur = None                                                       # oracle outputs the context's planes must equal
    trk = None                                                       # tracker model: dict(sw, sb, window) once begun
    side = torch.cuda.Stream()
    ctx = api.Context(spec, aux=aux)
    # the ingest pipe of the same context, used in between (include/slx.h: the context's own frames / outputs are not touched)
    pipe, pipe_slots, pipe_sets = None, int(rng.integers(2, 5)), int(rng.integers(1, 4))
    pipe_free, pipe_acquired, pipe_queue = 0, None, []               # model: free slots, the acquired slot's buffer, submitted slots' references
    primary = "z"
    try:
        if rng.random() < 0.5:
            expect_error(lambda: ctx.decode(), "decode without frames")
        expect_error(lambda: ctx.get_output("z"), "get_output before any decode")
        n_ops = int(rng.integers(8, 40))
        for _ in range(n_ops):
            op = int(rng.integers(0, 16))
            have_all = all(f is not None for f in frames["p"]) and all(f is not None for f in frames["g"])
            if op <= 2:                                              # a frame, or all of them
                targets = [(g, i) for g in ("p", "g") for i in range(len(frames[g]))]
                if not targets:
                    continue
                if op == 0 or not have_all:
                    chosen = targets if not have_all or rng.random() < 0.3 else [targets[int(rng.integers(0, len(targets)))]]
                else:
                    chosen = [targets[int(rng.integers(0, len(targets)))]]
                for g, i in chosen:
                    img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
                    how = int(rng.integers(0, 3))
                    grp = api.GROUP_PHASE if g == "p" else api.GROUP_GRAY
                    if how == 0:
                        ctx.set_frame(grp, i, img)
                    elif how == 1:
                        wide = np.zeros((h, w + 12), dtype=np.uint8)
                        wide[:, :w] = img
                        ctx.set_frame(grp, i, wide[:, :w])
                    else:
                        pitch = w + int(rng.choice([0, 4, 64]))
                        dev = torch.zeros((h, pitch), dtype=torch.uint8, device="cuda")
                        dev[:, :w] = torch.from_numpy(img).cuda()
                        torch.cuda.synchronize()
                        keep.append(dev)
                        ctx.set_frame(grp, i, dev[:, :w])
                    frames[g][i] = img
                say("set_frame x%d" % len(chosen))
                if rng.random() < 0.2:
                    expect_error(lambda: ctx.set_frame(api.GROUP_PHASE, n_phase + 3, np.zeros((h, w), dtype=np.uint8)), "a frame index past the stack")
            elif op == 3 or op == 4:                                 # decode
                if not have_all:
                    expect_error(lambda: ctx.decode(), "decode with frames missing")
                    continue
                use_side = rng.random() < 0.3
                ctx.decode(stream=side.cuda_stream if use_side else None)
                say("decode", "side" if use_side else "own")
                ph = np.stack(frames["p"]) if n_phase else None
                gr = np.stack(frames["g"]) if n_gray else None
                cur = O.pipeline(spec, ph, gr, want=tuple(planes))
                trk = None
            elif op == 5 and cur is not None:                        # read planes back
                for p in planes:
                    if rng.random() < 0.6:
                        same("get_output(%s)" % p, ctx.get_output(p), cur[p])
                say("get_output")
                bogus = [p for p in ("U", "x", "k", "mask", "pix") if p not in planes and not (p == "pix" and mode < 2)]
                if bogus and rng.random() < 0.3:
                    expect_error(lambda: ctx.get_output(bogus[0]), "a plane that was not enabled (%s)" % bogus[0])
            elif op == 6 and cur is not None:
                want_cloud = O.point_cloud(spec, cur["z"])
                same("point cloud", ctx.get_point_cloud() if rng.random() < 0.5 else np.array(ctx.get_point_cloud_view()), want_cloud)
                say("cloud")
                if rng.random() < 0.4:                               # ... and its text, formatted on the device (round 5)
                    mag = np.abs(want_cloud)
                    in_range = bool(np.all(np.isfinite(want_cloud)) and np.all((mag == 0) | ((mag >= 1e-5) & (mag < 1e15))))
                    try:
                        text, n_pts = ctx.get_point_cloud_text()
                        if not in_range or n_pts != len(want_cloud) or text != ("".join("%g %g %g\n" % tuple(q) for q in want_cloud)).encode():
                            raise Mismatch("the point cloud's text differs")
                    except api.SlxError as e:
                        if in_range or e.code != api.ERR_UNAVAILABLE:
                            raise Mismatch("the point cloud's text: %s" % str(e)[:80])
                    say("text")
            elif op == 7:                                            # a batch decode into the caller's buffers, context planes untouched
                n = int(rng.integers(1, 4))
                bph = rng.integers(0, 256, size=(n, n_phase, h, w), dtype=np.uint8) if n_phase else None
                bgr = rng.integers(0, 256, size=(n, n_gray, h, w), dtype=np.uint8) if n_gray else None
                tph = torch.from_numpy(bph).cuda() if n_phase else None
                tgr = torch.from_numpy(bgr).cuda() if n_gray else None
                z = torch.full((n, h, w), -3.0, dtype=torch.float64, device="cuda")
                torch.cuda.synchronize()
                use_side = rng.random() < 0.3
                ctx.decode_batch(n, tph, tgr, z, stream=side.cuda_stream if use_side else None)
                ctx.synchronize()
                torch.cuda.synchronize()
                for s in range(n):
                    ref = O.pipeline(spec, None if bph is None else bph[s], None if bgr is None else bgr[s], want=("z",))["z"]
                    same("batch z[%d]" % s, z[s].cpu().numpy(), ref)
                if rng.random() < 0.5:
                    same("cloud of a batch plane", ctx.point_cloud_of_depth(z[n - 1]), O.point_cloud(spec, ref))
                say("decode_batch x%d" % n, "side" if use_side else "own")
            elif op == 8 and cur is not None and "U" in planes:      # tracker
                img = rng.integers(0, 256, size=(h, w), dtype=np.uint8) if rng.random() < 0.3 else \
                    np.clip(128 + 100 * np.sign(np.sin((np.arange(w)[None, :] + rng.uniform(0, 20) + 0.05 * np.arange(h)[:, None]) / rng.uniform(1.5, 5))) +
                            rng.normal(0, 5, (h, w)), 0, 255).astype(np.uint8)
                if trk is None or rng.random() < 0.15:
                    window = 21 if rng.random() < 0.6 else int(rng.integers(1, 12)) * 2 + 1
                    if rng.random() < 0.2:
                        expect_error(lambda: ctx.track_begin(img, window=window + 1), "an even tracker window")
                    ctx.track_begin(img, window=window)
                    sw, sb = O.strip_regression(img, window)
                    trk = {"sw": sw, "sb": sb, "window": window}
                    say("track_begin", window)
                else:
                    feed = int(rng.integers(0, 4))
                    if feed == 0:
                        ctx.track_next(img)
                    elif feed == 1:
                        buf = ctx.track_image_buffer()
                        buf[:] = img
                        ctx.track_next(buf)
                    elif feed == 2:
                        dev = torch.from_numpy(img).cuda()
                        torch.cuda.synchronize()
                        keep.append(dev)
                        ctx.track_next(dev)
                    else:
                        ctx.track_next_batch(img[None])
                    sw1, sb1 = O.strip_regression(img, trk["window"])
                    dP = O.delta_p(trk["sw"], trk["sb"], sw1, sb1)
                    U = cur["U"] + dP.astype(np.float64)
                    tri = O.triangulate(spec, U, want=("z", "x", "y"))
                    cur = dict(cur, U=U, z=tri["z"], x=tri["x"], y=tri["y"])
                    trk.update(sw=sw1, sb=sb1)
                    say("track_next feed", feed)
                    same("deltaP", ctx.get_output("deltaP"), dP)
                same("stripW", ctx.get_output("stripW"), trk["sw"])
            elif op == 8 and cur is None and "U" in planes:
                expect_error(lambda: ctx.track_next(np.zeros((h, w), dtype=np.uint8)), "track_next before track_begin")
            elif op == 9:
                v = int(rng.choice([0, 0, 1, 3]))
                ctx.set_variant(v)
                say("variant", v)
            elif op == 10:
                key = str(rng.choice(["strip_rows", "weave", "tiers", "strip_waves", "tail_pct", "stream", "stream_rows"]))
                kv = {key: int(rng.integers(0, 3 if key == "stream" else 5))}
                ctx.set_tuning(**kv)
                if rng.random() < 0.2:
                    expect_error(lambda: ctx.set_tuning(stream=7), "a tuning value out of range")
                say("tuning", kv)
            elif op == 11:
                if rng.random() < 0.5:
                    ctx.synchronize()
                else:
                    ctx.enable_timing(bool(rng.random() < 0.5))
            elif op >= 12:                                           # the ingest pipe
                if pipe is None:
                    host_result = bool(rng.random() < 0.7)
                    pipe = api.Pipe(ctx, slots=pipe_slots, sets_per_slot=pipe_sets, host_result=host_result)
                    pipe_free = pipe_slots
                    say("pipe", pipe_slots, pipe_sets, host_result)
                    expect_error(lambda: pipe.collect(), "collect with nothing submitted")
                    expect_error(lambda: pipe.submit(), "submit without an acquired slot")
                    continue
                what = int(rng.integers(0, 3))
                if what == 0 and pipe_acquired is None:
                    if pipe_free == 0:
                        expect_error(lambda: pipe.acquire(), "acquire with every slot in flight")
                        continue
                    buf = pipe.acquire()
                    pipe_free -= 1
                    n = int(rng.integers(1, pipe_sets + 1))
                    data = rng.integers(0, 256, size=(n, pipe.n_planes, h, w), dtype=np.uint8)
                    buf[:n, :, :, :w] = data
                    pipe_acquired = data
                    say("pipe.acquire", n)
                elif what == 1 and pipe_acquired is not None:
                    n = pipe_acquired.shape[0]
                    if rng.random() < 0.1:
                        expect_error(lambda: pipe.submit(pipe_sets + 1), "more frame-sets than the slot holds")
                    pipe.submit(n)
                    refs = [O.pipeline(spec, pipe_acquired[s_, :n_phase] if n_phase else None, pipe_acquired[s_, n_phase:] if n_gray else None, want=(primary,))[primary]
                            for s_ in range(n)]
                    pipe_queue.append(refs)
                    pipe_acquired = None
                    say("pipe.submit", n)
                elif what == 2 and pipe_queue:
                    refs = pipe_queue.pop(0)
                    got = pipe.collect()
                    if pipe.host_result:
                        if got.shape[0] != len(refs):
                            raise Mismatch("pipe.collect gave %d frame-sets, %d were submitted" % (got.shape[0], len(refs)))
                        for s_ in range(len(refs)):
                            same("pipe result %d" % s_, got[s_], refs[s_])
                    else:
                        addr, n = got
                        if n != len(refs):
                            raise Mismatch("pipe.collect gave %d frame-sets, %d were submitted" % (n, len(refs)))
                        dev = torch.empty((n, h, w), dtype=torch.float64, device="cuda")
                        torch.cuda.synchronize()
                        import ctypes
                        hip = ctypes.CDLL("libamdhip64.so")
                        if hip.hipMemcpy(ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(addr), ctypes.c_size_t(n * h * w * 8), 3) != 0:
                            raise Mismatch("hipMemcpy of the pipe's device result failed")
                        for s_ in range(n):
                            same("pipe device result %d" % s_, dev[s_].cpu().numpy(), refs[s_])
                    pipe_free += 1
                    say("pipe.collect")
        if cur is not None:                                           # whatever happened: the planes at the end
            for p in planes:
                same("final get_output(%s)" % p, ctx.get_output(p), cur[p])
    except (Mismatch, api.SlxError) as e:
        return {"seed": seed, "config": name, "w": w, "h": h, "planes": planes, "what": "%s: %s" % (type(e).__name__, e), "last_calls": log[-12:]}
    finally:
        if pipe is not None and rng.random() < 0.7:                  # else: the context goes first (slx_destroy handles a live pipe)
            pipe.close()
        ctx.close()
    return None


t_end = time.time() + budget
i = failures = 0
while more(i):
    seed = seed0 * 100003 + i
    i += 1
    try:
        bad = one_case(seed)
    except Exception as e:
        bad = {"seed": seed, "what": "%s: %s" % (type(e).__name__, e)}
    if bad:
        failures += 1
        print(json.dumps(bad), flush=True)
        if failures >= 20:
            break
print("fuzz_api: %d contexts, %d failures" % (i, failures))
sys.exit(1 if failures else 0)
