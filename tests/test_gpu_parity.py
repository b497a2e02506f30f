"""GPU parity tests: the HIP path, called through the C ABI (libslx.so), against the CPU
oracle on the same inputs.  Bar (BASELINE.json north_star): depth within 1e-4 mm RMS of the
CPU path (1e-5 for the 4-frequency x 8-step case), phase indices bit-exact.  The kernel is
written to round exactly like the oracle, so these tests demand bit-equality everywhere and
state the contractual tolerance next to it.

Oracle = oracle/ (CPU restatement, "parity unpinned": the reference has no tests and is
unbuildable here).  Nothing in this file reads /root/reference.
"""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RMS_TOL_MM = 1e-4
RMS_TOL_MM_C5 = 1e-5


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def small_spec(synth, name, w=64, h=48):
    spec = dict(synth.make_spec(name))
    spec["width"], spec["height"] = w, h
    spec["calib"] = synth.scaled_calibration(w, h, spec["proj_width"])
    return spec


def exhaustive_planes(width=511):
    d = np.arange(-255, 256)
    d02, d13 = np.meshgrid(d, d, indexing="ij")
    p = np.zeros((4, 511, width), dtype=np.uint8)
    p[0, :, :511] = np.maximum(d02, 0)
    p[2, :, :511] = np.maximum(-d02, 0)
    p[1, :, :511] = np.maximum(d13, 0)
    p[3, :, :511] = np.maximum(-d13, 0)
    return p


def all_outputs(spec):
    mode = spec["mode"]
    want = ["z", "x", "y", "U", "pix"]
    if spec.get("gray_bits"):
        want.append("gray")
    if mode in (3, 4):
        want.append("mask")
        if spec["n_freq"] > 1:
            want.append("k")
    return tuple(want)


def strip_outputs(spec):
    """What the strip kernel produces: the depth-side planes (the per-frequency pix planes and the Gray plane are the generic
    kernel's)."""
    return tuple(w for w in all_outputs(spec) if w not in ("pix", "gray"))


def strip_kernel_name(spec, aux, gray_on_ring=True):
    """The slx_strip_kernel instantiation a launch of this configuration is MEANT to run as, written as slx_last_kernel (and
    rocprofv3) print it: <mode, frequencies, Gray bits riding the DMA ring, steps, optional planes>.  Six Gray bits ride the ring
    (the reference's and configuration 3's count); a launch that fell back to ordinary Gray loads reports 0 there and fails the
    assertion that uses this name -- round 3's silent REF regression was exactly that fall."""
    mode = spec["mode"]
    F = 1 if mode == 2 else spec["n_freq"]
    gb = 6 if (gray_on_ring and mode in (2, 4) and spec.get("gray_bits") == 6) else 0
    return "slx_strip_kernel<%d, %d, %d, %d, %s>" % (mode, F, gb, spec.get("n_steps", 4), "true" if aux else "false")


def assert_same(got, want, names, tol=RMS_TOL_MM):
    for n in names:
        g, w = got[n], want[n]
        assert g.shape == w.shape, n
        if n == "z":
            both = np.isfinite(g) & np.isfinite(w)
            rms = float(np.sqrt(np.mean((g[both] - w[both]) ** 2))) if both.any() else 0.0
            assert rms <= tol, "depth RMS %.3e mm exceeds the contractual %.0e" % (rms, tol)
        assert np.array_equal(g, w, equal_nan=True), "%s differs in %d of %d elements" % (
            n, int(np.sum(~((g == w) | (np.isnan(g) & np.isnan(w))))), g.size)


# ------------------------------------------------------------------ wrapped phase, exhaustive
@pytest.mark.parametrize("width", [511, 512])
def test_wrapped_phase_exhaustive(api, oracle, synth, golden_dir, width):
    """All 511 x 511 possible (I0-I2, I1-I3) inputs of CDecodePhase::CountResult, every period the
    configurations use.  width 511 takes the byte-granular path, 512 the dword path."""
    tables = json.load(open(os.path.join(golden_dir, "wrapped_phase_tables.json")))["tables"]
    planes = exhaustive_planes(width)
    for T in (40, 20, 30, 160, 240, 1280, 1920, 4096, 8, 64, 512, 7, 1000003):
        spec = {"width": width, "height": 511, "mode": synth.MODE_PHASE_ONLY, "n_freq": 1, "n_steps": 4, "periods": [T]}
        got = api.decode_frameset(spec, planes, None, want=("pix",))["pix"][0]
        ref = oracle.pipeline(spec, planes, None, want=("pix",))["pix"][0]
        assert np.array_equal(got, ref), "T=%d: %d mismatches" % (T, int((got != ref).sum()))
        if str(T) in tables:
            sha = hashlib.sha256(np.ascontiguousarray(got[:, :511]).tobytes()).hexdigest()
            assert sha == tables[str(T)]["sha256"]


def test_wrapped_phase_random_bytes_nstep(api, oracle, synth):
    """x1: N-step for N != 4 (literal float/double path) and N == 4 on unstructured bytes."""
    for n_steps in (3, 4, 5, 8, 16):
        for T in (40, 512):
            spec = {"width": 333, "height": 65, "mode": synth.MODE_PHASE_ONLY, "n_freq": 1, "n_steps": n_steps, "periods": [T]}
            ph, _ = synth.random_planes(spec, seed=n_steps * 10 + T)
            got = api.decode_frameset(spec, ph, None, want=("pix",))["pix"]
            ref = oracle.pipeline(spec, ph, None, want=("pix",))["pix"]
            assert np.array_equal(got, ref), (n_steps, T)


# ------------------------------------------------------------------ Gray decode alone
@pytest.mark.parametrize("bits,width", [(6, 640), (1, 17), (10, 96), (16, 64)])
def test_gray_only(api, oracle, synth, bits, width):
    spec = {"width": width, "height": 37, "mode": synth.MODE_GRAY_ONLY, "gray_bits": bits,
            "gray_stripe": max(1, 65536 // (1 << bits)) if bits == 16 else max(1, 1280 // (1 << bits)),
            "gray_lut": synth.standard_gray_lut(bits) if bits < 16 else (np.arange(1 << 16) % 30000).astype(np.int16)}
    _, gr = synth.random_planes(spec, seed=bits)
    gr[:, :, : width // 3] = np.where(gr[:, :, : width // 3] > 127, 220, 20)     # also clean patterns and exact ties
    gr[1::2, :, : width // 6] = gr[0::2, :, : width // 6]
    got = api.decode_frameset(spec, None, gr, want=("gray",))["gray"]
    ref = oracle.pipeline(spec, None, gr, want=("gray",))["gray"]
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("name,shape,n_sets", [("C4", (516, 71), 9), ("C2", (320, 33), 40), ("C1", (192, 150), 16), ("C4", (1920, 150), 6)])
def test_stream_kernel_geometries_and_repeated_launches(api, oracle, synth, torch_cuda, name, shape, n_sets):
    """slx_stream_kernel (resident waves taking short items from per-column queues): every rows-per-item choice on ragged tiles
    against the oracle; the queue counters carry over from launch to launch of one geometry (three launches each) and are zeroed
    when the geometry changes (item length, frame-set count, back and forth); a launch on a caller's stream in between;
    slx_last_kernel says which kernel ran."""
    torch = torch_cuda
    spec = small_spec(synth, name, *shape)
    H, W = spec["height"], spec["width"]
    sets = [synth.random_planes(spec, seed=7000 + s)[0] for s in range(n_sets)]
    refs = [oracle.pipeline(spec, p, None, want=("z",))["z"] for p in sets]
    ph = torch.from_numpy(np.stack(sets)).cuda()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        ctx.set_variant(2)
        for rows, n in ((2, n_sets), (3, n_sets), (2, n_sets), (7, n_sets - 1), (16, 2), (0, n_sets), (4, n_sets)):
            ctx.set_tuning(stream=2, stream_rows=rows)
            for rep in range(3):
                z = torch.full((n, H, W), -7.0, dtype=torch.float64, device="cuda")
                torch.cuda.synchronize()
                ctx.decode_batch(n, ph[:n], None, z, stream=side.cuda_stream if rep == 1 else None)
                ctx.synchronize()
                torch.cuda.synchronize()
                assert ctx.last_kernel().startswith("slx_stream_kernel"), ctx.last_kernel()
                for k in range(n):
                    assert np.array_equal(z[k].cpu().numpy(), refs[k], equal_nan=True), (rows, n, rep, k)
        ctx.set_tuning(stream=1)
        z = torch.full((n_sets, H, W), -7.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ctx.decode_batch(n_sets, ph, None, z)
        ctx.synchronize()
        assert ctx.last_kernel().startswith("slx_strip_kernel")
        assert np.array_equal(z[0].cpu().numpy(), refs[0], equal_nan=True)


@pytest.mark.parametrize("shape,n_sets,row_offset,std_lut", [((320, 37), 7, 0, True), ((1280, 30), 3, 11, True), ((132, 65), 5, 0, True), ((320, 41), 4, 0, False),
                                                             ((1280, 1024), 8, 0, True)])
def test_gray_stream_kernel_geometries_and_repeated_launches(api, oracle, synth, torch_cuda, shape, n_sets, row_offset, std_lut):
    """slx_gstream_kernel (round 6: the stream kernel of the reference's own mode, 6 Gray bits on the DMA ring + 4 steps; a row is two
    ring chunks there): every rows-per-item choice (its default is ONE row) on ragged tiles against the oracle; queue counters carried
    over three launches of a geometry and zeroed when it changes; a launch on a caller's stream in between; noise-free stripes, exact
    ties and random bytes in the Gray planes; a table that is not the reflected code; the reference's real size, where the planner
    takes the kernel by itself from 7 frame-sets on; slx_last_kernel names the kernel."""
    torch = torch_cuda
    spec = small_spec(synth, "REF", *shape) if shape != (1280, 1024) else dict(synth.make_spec("REF"))
    spec["row_offset"] = row_offset
    if not std_lut:
        spec["gray_lut"] = ((np.arange(64) * 5 + 3) % 64).astype(np.int16)      # any table: lut[gray] = bin is looked up, not computed
    H, W = spec["height"], spec["width"]
    sets, grays = [], []
    for s in range(n_sets):
        ph, gr = synth.random_planes(spec, seed=8100 + s)
        gr[:, :, : W // 3] = np.where(gr[:, :, : W // 3] > 127, 220, 20)          # clean patterns ...
        gr[1::2, :, : W // 6] = gr[0::2, :, : W // 6]                              # ... and exact ties (pattern == inverse -> bit 0)
        sets.append(ph)
        grays.append(gr)
    refs = [oracle.pipeline(spec, p, g, want=("z",), threads=8)["z"] for p, g in zip(sets, grays)]
    ph = torch.from_numpy(np.stack(sets)).cuda()
    gr = torch.from_numpy(np.stack(grays)).cuda()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        ctx.set_variant(2)
        for rows, n in ((0, n_sets), (2, n_sets), (1, n_sets), (3, n_sets), (0, n_sets), (7, n_sets - 1), (16, 2), (4, n_sets)):
            ctx.set_tuning(stream=2, stream_rows=rows)
            for rep in range(3):
                z = torch.full((n, H, W), -7.0, dtype=torch.float64, device="cuda")
                torch.cuda.synchronize()
                ctx.decode_batch(n, ph[:n], gr[:n], z, stream=side.cuda_stream if rep == 1 else None)
                ctx.synchronize()
                torch.cuda.synchronize()
                assert ctx.last_kernel() == "slx_gstream_kernel: resident waves, %d-row items from queues" % (rows or 1), ctx.last_kernel()
                for k in range(n):
                    assert np.array_equal(z[k].cpu().numpy(), refs[k], equal_nan=True), (rows, n, rep, k)
        # the planner's own choice: the stream kernel for a launch of >= 8 items per resident wave (the reference's size: 7 frame-sets), else the strip kernel
        ctx.set_tuning(stream=0, stream_rows=0)
        z = torch.full((n_sets, H, W), -7.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ctx.decode_batch(n_sets, ph, gr, z)
        ctx.synchronize()
        assert ctx.last_kernel().startswith("slx_gstream_kernel:" if shape == (1280, 1024) else "slx_strip_kernel<2, 1, 6, 4, false>:"), ctx.last_kernel()
        for k in range(n_sets):
            assert np.array_equal(z[k].cpu().numpy(), refs[k], equal_nan=True), k
        ctx.set_tuning(stream=1)
        ctx.decode_batch(n_sets, ph, gr, z)
        ctx.synchronize()
        assert ctx.last_kernel().startswith("slx_strip_kernel<2, 1, 6, 4, false>:")
        for k in range(n_sets):
            assert np.array_equal(z[k].cpu().numpy(), refs[k], equal_nan=True), k


def test_gray_and_phase_groups_far_apart_in_memory(api, oracle, synth, torch_cuda):
    """The two plane groups of a batch are separate allocations and may sit anywhere: here more than 2 GiB apart, in either order
    (the Gray planes ride the DMA ring through a descriptor of their own; round 3's single descriptor made such a launch fall back
    to ordinary Gray loads).  REF and the Gray-mask mode, strip kernel, against the oracle."""
    torch = torch_cuda
    for name in ("REF", "C3"):
        spec = small_spec(synth, name, 256, 40)
        n_sets = 3
        sets = [synth.random_planes(spec, seed=600 + s) for s in range(n_sets)]
        refs = [oracle.pipeline(spec, p, g, want=("z",))["z"] for p, g in sets]
        pnp, gnp = np.stack([p for p, _ in sets]), np.stack([g for _, g in sets])
        for gray_first in (False, True):
            # one 2.5 GiB buffer, one group at either end of it: the distance is certain, whatever the allocator does
            arena = torch.empty(((5 << 29) + (1 << 24),), dtype=torch.uint8, device="cuda")
            lo_np, hi_np = (gnp, pnp) if gray_first else (pnp, gnp)
            lo = arena[: lo_np.size].view(lo_np.shape)
            hi = arena[arena.numel() - (1 << 24): arena.numel() - (1 << 24) + hi_np.size].view(hi_np.shape)
            lo.copy_(torch.from_numpy(lo_np))
            hi.copy_(torch.from_numpy(hi_np))
            ph, gr = (hi, lo) if gray_first else (lo, hi)
            assert abs(ph.data_ptr() - gr.data_ptr()) > (1 << 31)
            z = torch.full((n_sets, spec["height"], spec["width"]), -7.0, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            with api.Context(spec) as ctx:
                ctx.set_variant(2)
                ctx.decode_batch(n_sets, ph, gr, z)
                ctx.synchronize()
                assert ctx.last_kernel().startswith(strip_kernel_name(spec, aux=False) + ":"), ctx.last_kernel()   # Gray planes on the ring, not <..., 0, 4, ...>
            for s in range(n_sets):
                assert np.array_equal(z[s].cpu().numpy(), refs[s], equal_nan=True), (name, gray_first, s)
            del arena, lo, hi, ph, gr


@pytest.mark.parametrize("variant", [2, 1, 0])
def test_decoder_objects_on_the_strip_path(api, oracle, synth, torch_cuda, variant):
    """CDecodePhase::Decode and CDecodeGray::Decode alone (modes PHASE_ONLY / GRAY_ONLY) as batches through slx_decode_batch:
    variant 2 must take slx_decoder_strip_kernel (DMA ring, 4 / 12 planes), variant 1 the generic kernel; every rows-per-item
    choice, ragged heights, a pitch in the planes, unstructured bytes plus clean patterns and exact ties; against the oracle."""
    torch = torch_cuda
    rng = np.random.default_rng(1234)
    for W, H, pitch in ((512, 71, 512), (1280, 33, 1344), (196, 129, 196)):
        n_sets = 3
        # phase decoder: all 511 x 511 difference pairs in the first rows, unstructured bytes below
        pspec = {"width": W, "height": H, "mode": synth.MODE_PHASE_ONLY, "n_freq": 1, "n_steps": 4, "periods": [40]}
        ph = rng.integers(0, 256, size=(n_sets, 4, H, W), dtype=np.uint8)
        ex = exhaustive_planes(511)[:, : min(H, 511), : min(W, 511)]
        ph[0, :, : ex.shape[1], : ex.shape[2]] = ex
        gspec = {"width": W, "height": H, "mode": synth.MODE_GRAY_ONLY, "gray_bits": 6, "gray_stripe": 20, "gray_lut": synth.standard_gray_lut(6)}
        gr = rng.integers(0, 256, size=(n_sets, 12, H, W), dtype=np.uint8)
        gr[:, :, :, : W // 3] = np.where(gr[:, :, :, : W // 3] > 127, 220, 20)
        gr[:, 1::2, :, : W // 6] = gr[:, 0::2, :, : W // 6]                    # exact ties: bit 0
        for spec, planes, key in ((pspec, ph, "pix"), (gspec, gr, "gray")):
            refs = [oracle.pipeline(spec, planes[s] if key == "pix" else None, planes[s] if key == "gray" else None, want=(key,))[key].reshape(H, W)
                    for s in range(n_sets)]
            dev = torch.zeros((n_sets, planes.shape[1], H, pitch), dtype=torch.uint8, device="cuda")
            dev[..., :W] = torch.from_numpy(planes).cuda()
            view = dev[..., :W]
            for rows in (0, 1, 2, 3, 5, 16):
                out = torch.full((n_sets, H, W), -7.0, dtype=torch.float64, device="cuda")
                torch.cuda.synchronize()
                with api.Context(spec) as ctx:
                    ctx.set_variant(variant)
                    ctx.set_tuning(strip_rows=rows)
                    ctx.decode_batch(n_sets, view if key == "pix" else None, view if key == "gray" else None, out, row_stride=pitch)
                    ctx.synchronize()
                    meant = "slx_fused_kernel<%d, " % spec["mode"] if variant == 1 else "slx_decoder_strip_kernel<%d>:" % spec["mode"]
                    assert ctx.last_kernel().startswith(meant), (ctx.last_kernel(), meant)
                for s in range(n_sets):
                    assert np.array_equal(out[s].cpu().numpy(), refs[s]), (W, H, key, variant, rows, s)
    # what the strip variant refuses stays refused (and the automatic choice falls back to the generic kernel)
    odd = {"width": 64, "height": 8, "mode": synth.MODE_GRAY_ONLY, "gray_bits": 5, "gray_stripe": 40, "gray_lut": synth.standard_gray_lut(5)}
    g5 = torch.zeros((1, 10, 8, 64), dtype=torch.uint8, device="cuda")
    o5 = torch.zeros((1, 8, 64), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with api.Context(odd) as ctx:
        ctx.set_variant(variant)
        if variant == 2:
            with pytest.raises(api.SlxError) as e:
                ctx.decode_batch(1, None, g5, o5)
            assert e.value.code == api.ERR_UNAVAILABLE
        else:
            ctx.decode_batch(1, None, g5, o5)
            ctx.synchronize()


# ------------------------------------------------------------------ committed fixtures
@pytest.mark.parametrize("name", ["C1x4", "C2", "C3", "C5"])
def test_scene_fixtures(api, synth, golden_dir, name):
    d = np.load(os.path.join(golden_dir, "scene_%s.npz" % name))
    spec = small_spec(synth, name)
    want = tuple(k[4:] for k in d.files if k.startswith("out_"))
    got = api.decode_frameset(spec, d["phase"] if "phase" in d.files else None,
                              d["gray_planes"] if "gray_planes" in d.files else None, want=want)
    assert_same(got, {w: d["out_" + w] for w in want}, want, RMS_TOL_MM_C5 if name == "C5" else RMS_TOL_MM)


# ------------------------------------------------------------------ BASELINE configurations, full size
@pytest.mark.parametrize("name,scene", [("C1", "tilted"), ("C1x4", "sphere"), ("REF", "sphere"), ("C2", "sphere"),
                                        ("C3", "sphere"), ("C4", "tilted")])
def test_baseline_configs_full_size(api, oracle, synth, name, scene):
    spec = synth.make_spec(name)
    ph, gr, _ = synth.render(spec, scene, seed=0x5EED + len(name), noise_sigma=2.0)
    want = all_outputs(spec)
    info = {}
    got = api.decode_frameset(spec, ph, gr, want=want, info=info)
    assert info["kernel"].startswith("slx_fused_kernel<%d, " % spec["mode"]), info   # the per-frequency pix planes / the Gray plane: the generic kernel's
    ref = oracle.pipeline(spec, ph, gr, want=want, threads=8)
    assert_same(got, ref, want)
    assert (got["z"] > 0).mean() > 0.9
    # x, y, U, the fringe orders and the mask from the strip kernel (variant 2 refuses to fall back), Gray planes on the DMA ring
    sw = strip_outputs(spec)
    got = api.decode_frameset(spec, ph, gr, want=sw, variant=api.VARIANT_STRIP, info=info)
    assert info["kernel"].startswith(strip_kernel_name(spec, aux=True) + ":"), info
    assert_same(got, ref, sw)
    # the call the reference's host loop makes -- one frame-set, depth only, automatic plan: the same kernel family, no optional planes
    got = api.decode_frameset(spec, ph, gr, want=("z",), info=info)
    assert info["kernel"].startswith(strip_kernel_name(spec, aux=False) + ":"), info
    assert_same(got, ref, ("z",))


def test_baseline_config_c5(api, oracle, synth):
    """4096 x 3000, 4-frequency x 8-step, tolerance 1e-5 mm RMS (bit-exact in fact): the generic kernel with every
    output, then the depth of the literal-arithmetic kernel and of the 8-step strip kernel."""
    spec = synth.make_spec("C5")
    ph, _, _ = synth.render(spec, "tilted", seed=0x5EED + 5, noise_sigma=1.0)
    ref = oracle.pipeline(spec, ph, None, want=("z", "k", "U"), threads=8)
    got = api.decode_frameset(spec, ph, None, want=("z", "k", "U"))
    assert_same(got, ref, ("z", "k", "U"), RMS_TOL_MM_C5)
    info = {}
    for variant, kernel in ((api.VARIANT_GENERIC, "slx_fused_kernel<3, 4, false, false>"), (api.VARIANT_STRIP, "slx_strip_kernel<3, 4, 0, 8, false>:"),
                            (api.VARIANT_AUTO, "slx_strip_kernel<3, 4, 0, 8, false>:")):
        got = api.decode_frameset(spec, ph, None, want=("z",), variant=variant, info=info)
        assert info["kernel"].startswith(kernel), (variant, info)
        assert_same(got, ref, ("z",), RMS_TOL_MM_C5)
    got = api.decode_frameset(spec, ph, None, want=("z", "k", "U"), variant=api.VARIANT_STRIP, info=info)
    assert info["kernel"].startswith("slx_strip_kernel<3, 4, 0, 8, true>:"), info
    assert_same(got, ref, ("z", "k", "U"), RMS_TOL_MM_C5)


# ------------------------------------------------------------------ unstructured inputs, every branch
@pytest.mark.parametrize("name", ["C1", "C1x4", "C2", "C3", "C5"])
@pytest.mark.parametrize("shape", [(48, 64), (31, 250), (9, 511), (3, 1001), (1, 1), (2, 3), (5, 130)])
@pytest.mark.parametrize("variant", [0, 1])
def test_random_bytes_all_outputs(api, oracle, synth, name, shape, variant):
    h, w = shape
    spec = small_spec(synth, name, w, h)
    ph, gr = synth.random_planes(spec, seed=h * 7919 + w)
    if gr is not None and w > 8:
        gr[:, :, : w // 2] = np.where(gr[:, :, : w // 2] > 127, 220, 20)
    want = all_outputs(spec)
    got = api.decode_frameset(spec, ph, gr, want=want, variant=variant)   # 0: cheap exact arithmetic, 1: literal
    ref = oracle.pipeline(spec, ph, gr, want=want)
    assert_same(got, ref, want)


@pytest.mark.parametrize("variant", [1, 2])
def test_mask_halo_crosses_wave_boundaries(api, oracle, synth, variant):
    """x3's 3-tap AND: invalid pixels planted on every quad / wave seam of a wide row (generic kernel: shuffles,
    62-quad waves; strip kernel: DPP wave shifts, 62-quad chunks of interleaved rows)."""
    spec = small_spec(synth, "C3", 1920, 6)
    ph, gr, _ = synth.render(spec, "tilted")
    rng = np.random.default_rng(3)
    cols = sorted(set([0, 1, 3, 4, 5, 243, 244, 247, 248, 249, 250, 251, 252, 255, 256, 491, 492, 495, 496, 1916, 1917, 1918, 1919]
                      + [int(c) for c in rng.integers(0, 1920, 60)]))
    for r in range(6):
        for c in cols[r::3]:
            for b in range(6):
                gr[2 * b, r, c], gr[2 * b + 1, r, c] = gr[2 * b + 1, r, c], gr[2 * b, r, c]
    ref = oracle.pipeline(spec, ph, gr, want=("mask", "z"))
    assert ref["mask"].min() == 0 and ref["mask"].max() == 1
    if variant == 1:
        got = api.decode_frameset(spec, ph, gr, want=("mask", "z"), variant=variant)
        assert_same(got, ref, ("mask", "z"))
    else:
        got = api.decode_frameset(spec, ph, gr, want=("z",), variant=variant)
        assert_same(got, ref, ("z",))
        assert np.all(got["z"][ref["mask"] == 0] == 0)


# ------------------------------------------------------------------ kernel variants (fast paths)
def quadrant_tie_planes(spec, reps=8):
    """Every combination of quarter-turn phases across the frequencies, several amplitudes: these make
    (U_{f-1} - pix_f)/T_f + 0.5 land exactly on integers (rounding ties of the temporal unwrap)."""
    F, H, W = spec["n_freq"], spec["height"], spec["width"]
    planes = np.zeros((F * 4, H, W), dtype=np.uint8)
    combos = 4 ** F
    idx = np.arange(H * W).reshape(H, W)
    amp = 20 + 30 * ((idx // combos) % reps)
    base = 128
    for f in range(F):
        quad = (idx // (4 ** f)) % 4                     # phase = quad * 90 degrees
        s = np.array([0, 1, 0, -1])[quad] * amp          # I0 - I2 ~ 2 sin
        c = np.array([1, 0, -1, 0])[quad] * amp          # I1 - I3 ~ 2 cos
        planes[f * 4 + 0] = base + s // 2
        planes[f * 4 + 2] = base - s // 2
        planes[f * 4 + 1] = base + c // 2
        planes[f * 4 + 3] = base - c // 2
    return planes


@pytest.mark.parametrize("name", ["C2", "C4", "C5x4"])
def test_unwrap_rounding_ties(api, oracle, synth, name):
    if name == "C5x4":
        spec = dict(small_spec(synth, "C5", 256, 16), n_steps=4)
    else:
        spec = small_spec(synth, name, 256, 16)
    spec["fov_min"], spec["fov_max"] = -1e300, 1e300
    ph = quadrant_tie_planes(spec)
    ref = oracle.pipeline(spec, ph, None, want=("z", "k", "U", "pix"))
    # the construction does hit ties: (U_prev - pix)/T + 0.5 is an exact integer somewhere
    T = spec["periods"]
    r = (ref["pix"][0] - ref["pix"][1]) / T[1] + 0.5
    assert np.any(r == np.floor(r))
    got = api.decode_frameset(spec, ph, None, want=("z", "k", "U"), variant=api.VARIANT_GENERIC)
    assert_same(got, ref, ("z", "k", "U"))
    got = api.decode_frameset(spec, ph, None, want=("z", "k", "U"), variant=api.VARIANT_GENERIC_FAST)
    assert_same(got, ref, ("z", "k", "U"))
    for variant in (api.VARIANT_AUTO, api.VARIANT_STRIP):
        got = api.decode_frameset(spec, ph, None, want=("z",), variant=variant)
        assert_same(got, ref, ("z",))
        got = api.decode_frameset(spec, ph, None, want=("z", "k", "U"), variant=variant)      # "phase indices bit-exact" on the fast path
        assert_same(got, ref, ("z", "k", "U"))


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_variants_exhaustive_wrapped_phase_through_depth(api, oracle, synth, variant):
    """The fast kernels only emit depth; with an unbounded FOV depth is a strictly monotonic function
    of pix, so depth parity over all 511 x 511 inputs checks their wrapped phase exhaustively
    (in-register SDWA byte differences, fast unwrap and in-range division included)."""
    planes = exhaustive_planes(512)
    for T in (40, 30, 240, 1920, 4096, 16384):
        spec = dict(synth.make_spec("C1"), width=512, height=511, periods=[T])
        spec["calib"] = synth.scaled_calibration(512, 511, 1280)
        spec["fov_min"], spec["fov_max"] = -1e300, 1e300
        ref = oracle.pipeline(spec, planes, None, want=("z", "pix"))
        got = api.decode_frameset(spec, planes, None, want=("z",), variant=variant)
        assert np.array_equal(got["z"], ref["z"], equal_nan=True), (variant, T, int((got["z"] != ref["z"]).sum()))
        z, pix = ref["z"].ravel(), ref["pix"][0].ravel()     # distinct pix -> distinct z within a row
        row = 200 * 512
        pr, zr = pix[row:row + 511], z[row:row + 511]
        assert len(np.unique(np.round(zr, 12))) >= len(np.unique(pr)) * 0.99


@pytest.mark.parametrize("T,S", [(40, 20), (30, 15), (37, 18), (16384, 8192), (2, 1), (1001, 3)])
def test_gray_phase_merge_exhaustive(api, oracle, synth, torch_cuda, T, S):
    """The Gray / phase merge (a5, R/CCalculation.cpp:570-587) of the fast kernels is ONE rounding of the exact sum gray + phase + c
    (merge_gray_phase, csrc/slx_kernels.hip) where the reference chains exact double operations behind two comparisons against 0.25 T and
    0.75 T: every wrapped phase the 511 x 511 byte differences can produce, on even and on odd stripes (all 64 bins), for even / odd /
    large periods (odd T: the thresholds and T / 2 are not integers) -- projector column U and depth (unbounded FOV) against the oracle,
    from the strip kernel, the generic kernels and, as a batch, the stream kernel of this mode."""
    torch = torch_cuda
    spec = small_spec(synth, "C1x4", 512, 511)
    spec["periods"], spec["gray_stripe"] = [T], S
    spec["fov_min"], spec["fov_max"] = -1e300, 1e300
    ph = exhaustive_planes(512)
    rows, cols = np.meshgrid(np.arange(511), np.arange(512), indexing="ij")
    bins = (rows * 7 + cols // 8) % 64
    gray_code = bins ^ (bins >> 1)
    gr = np.empty((12, 511, 512), dtype=np.uint8)
    for b in range(6):
        bit = (gray_code >> b) & 1
        gr[2 * b] = np.where(bit == 1, 220, 20)
        gr[2 * b + 1] = np.where(bit == 1, 20, 220)
    ref = oracle.pipeline(spec, ph, gr, want=("z", "U"), threads=8)
    assert len(np.unique(ref["U"])) > 100000
    for variant in (api.VARIANT_STRIP, api.VARIANT_AUTO, api.VARIANT_GENERIC, api.VARIANT_GENERIC_FAST):
        got = api.decode_frameset(spec, ph, gr, want=("z", "U"), variant=variant)
        for w in ("U", "z"):
            assert np.array_equal(got[w], ref[w], equal_nan=True), (variant, w, int((got[w] != ref[w]).sum()))
    dph, dgr = torch.from_numpy(np.stack([ph, ph[:, ::-1].copy()])).cuda(), torch.from_numpy(np.stack([gr, gr[:, ::-1].copy()])).cuda()
    z = torch.full((2, 511, 512), -7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        ctx.set_tuning(stream=2)
        ctx.decode_batch(2, dph, dgr, z)
        ctx.synchronize()
        assert ctx.last_kernel().startswith("slx_gstream_kernel:"), ctx.last_kernel()
    assert np.array_equal(z[0].cpu().numpy(), ref["z"], equal_nan=True)
    ref2 = oracle.pipeline(spec, np.ascontiguousarray(ph[:, ::-1]), np.ascontiguousarray(gr[:, ::-1]), want=("z",), threads=8)["z"]   # set 1: the rows in reverse
    assert np.array_equal(z[1].cpu().numpy(), ref2, equal_nan=True)


@pytest.mark.parametrize("name,scene", [("C1", "sphere"), ("C1x4", "tilted"), ("REF", "tilted"), ("C2", "tilted"), ("C3", "sphere"), ("C4", "sphere")])
@pytest.mark.parametrize("variant", [1, 2, 3])
def test_variants_full_size(api, oracle, synth, name, scene, variant):
    spec = synth.make_spec(name)
    ph, gr, _ = synth.render(spec, scene, seed=11, noise_sigma=3.0)
    ref = oracle.pipeline(spec, ph, gr, want=("z",), threads=8)
    info = {}
    got = api.decode_frameset(spec, ph, gr, want=("z",), variant=variant, info=info)
    n4 = "true" if spec.get("n_steps", 4) == 4 else "false"
    meant = strip_kernel_name(spec, aux=False) + ":" if variant == 2 else "slx_fused_kernel<%d, %d, %s, false>" % (spec["mode"], 1 if spec["mode"] == 2 else spec["n_freq"], n4)
    assert info["kernel"].startswith(meant), (info, meant)
    assert_same(got, ref, ("z",))


@pytest.mark.parametrize("variant", [0, 2])
@pytest.mark.parametrize("shape", [(7, 64), (33, 1024), (130, 4096), (1, 4), (1200, 8), (37, 1920), (5, 500), (64, 20)])
def test_strip_kernel_geometries(api, oracle, synth, variant, shape):
    """Row bands, partial last bands, one-quad-wide and 1024-quad-wide strips, Gray + phase and 4-frequency."""
    h, w = shape
    for name in ("C1x4", "C5x4", "C3", "C5"):
        spec = small_spec(synth, "C5" if name == "C5x4" else name, w, h)
        if name != "C5":
            spec["n_steps"] = 4                                        # "C5": the 8-step x1 fast path
        ph, gr = synth.random_planes(spec, seed=h + w)
        if gr is not None and w >= 8:
            gr[:, :, : w // 2] = np.where(gr[:, :, : w // 2] > 127, 220, 20)
        ref = oracle.pipeline(spec, ph, gr, want=("z",))
        got = api.decode_frameset(spec, ph, gr, want=("z",), variant=variant)
        assert_same(got, ref, ("z",))


@pytest.mark.parametrize("periods", [[640], [640, 80], [1024, 128, 16], [4096, 512, 64, 8]])
@pytest.mark.parametrize("rows", [0, 1, 2, 3, 5, 16])
def test_eight_step_ring_depths(api, oracle, synth, torch_cuda, periods, rows):
    """The 8-step path moves one frequency per DMA chunk through the ring, with counted waits that depend on the chunk's place
    in the row and on the item's length.  Every frequency count (1 to 4 chunks per row) x item lengths from one row (the whole
    item is ring start-up and drain) to 16, depth and the optional planes, ragged tile."""
    torch = torch_cuda
    spec = small_spec(synth, "C5", 328, 43)
    spec["periods"], spec["n_freq"] = list(periods), len(periods)
    n_sets = 2
    sets = [synth.random_planes(spec, seed=300 + 7 * s + len(periods))[0] for s in range(n_sets)]
    want = ("z", "U", "x", "y") + (("k",) if len(periods) > 1 else ())
    refs = [oracle.pipeline(spec, p, None, want=want) for p in sets]
    ph = torch.from_numpy(np.stack(sets)).cuda()
    H, W = spec["height"], spec["width"]
    outs = {n: torch.full((n_sets, H, W), -7.0, dtype=torch.float64, device="cuda") for n in ("z", "U", "x", "y")}
    if len(periods) > 1:
        outs["k"] = torch.full((n_sets, len(periods) - 1, H, W), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        ctx.set_variant(2)
        ctx.set_tuning(strip_rows=rows)
        z_only = torch.full((n_sets, H, W), -7.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ctx.decode_batch(n_sets, ph, None, z_only)
        ctx.decode_batch_ex(n_sets, ph, None, **outs)
        ctx.synchronize()
    for s in range(n_sets):
        assert np.array_equal(z_only[s].cpu().numpy(), refs[s]["z"], equal_nan=True), (s, "z only")
        for n in want:
            assert np.array_equal(outs[n][s].cpu().numpy(), refs[s][n], equal_nan=True), (s, n)


def test_eight_step_sine_cosine_near_ties(api, oracle, synth, torch_cuda):
    """The 8-step fast path picks the octant with a saturating multiply, sat((|sy| - |sx|) 2^60): right as long as a non-zero
    difference is at least 2^-60.  It is: every term of the two sums is a multiple of 2^-24 (an 8-bit value, or RN(g * r) >= 0.7), so
    are the f32 sums, and after the 2/N scale any non-zero |sy| - |sx| is >= 2^-26.  This test sits on that edge: 8-tuples whose
    sine and cosine sums are equal in exact arithmetic (|sy| = |sx|: the 45 / 135 / 225 / 315 degree directions) -- exactly equal in f32
    when the odd steps are dark, and a few ulps apart when they are not (the two sums add the same terms in different orders) --
    against the oracle's literal compare, through U (= pix for one frequency) and z."""
    torch = torch_cuda
    rng = np.random.default_rng(808)
    W, H = 512, 96
    n = W * H
    g = rng.integers(0, 256, size=(8, n), dtype=np.int64)
    fam = rng.integers(0, 4, size=n)
    d = rng.integers(-255, 256, size=n)
    def pair(diff):                              # two bytes a, b with a - b = diff
        base = rng.integers(0, 256 - np.abs(diff))
        return np.where(diff >= 0, base + diff, base), np.where(diff >= 0, base, base - diff)
    # family 0: exact ties, odd steps dark: sy = g0 - g4 = D, sx = g2 - g6 = +-D
    # family 1: sy = sx in exact arithmetic (g3 = g7, any g1, g5): near ties of equal sign
    # family 2: sy = -sx in exact arithmetic (g1 = g5, any g3, g7)
    # family 3: unstructured
    sgn = np.where(rng.integers(0, 2, size=n) == 0, 1, -1)
    a0, a4 = pair(d)
    a2, a6 = pair(np.where(fam == 2, -d, np.where(fam == 0, sgn * d, d)))
    for k, v in ((0, a0), (4, a4), (2, a2), (6, a6)):
        g[k] = np.where(fam < 3, v, g[k])
    g[1] = np.where(fam == 0, 0, g[1]); g[3] = np.where(fam == 0, 0, g[3]); g[5] = np.where(fam == 0, 0, g[5]); g[7] = np.where(fam == 0, 0, g[7])
    g[7] = np.where(fam == 1, g[3], g[7])
    g[5] = np.where(fam == 2, g[1], g[5])
    planes = g.astype(np.uint8).reshape(8, H, W)
    for T in (1920, 37):
        spec = small_spec(synth, "C5", W, H)
        spec["periods"], spec["n_freq"] = [T], 1
        ref = oracle.pipeline(spec, planes, None, want=("z", "U"))
        # the construction does what it says: a good share of the pixels decode to the diagonal directions' neighbourhood
        frac = (ref["U"] - 0.5) / T
        near_diag = np.minimum.reduce([np.abs(frac - q) for q in (0.125, 0.375, 0.625, 0.875)]) < 1e-3
        assert near_diag.mean() > 0.5
        ph = torch.from_numpy(planes[None]).cuda()
        outs = {w: torch.full((1, H, W), -7.0, dtype=torch.float64, device="cuda") for w in ("z", "U")}
        torch.cuda.synchronize()
        for variant in (2, 1):                  # the strip kernel's 8-step path, and the generic kernel's literal arithmetic
            with api.Context(spec) as ctx:
                ctx.set_variant(variant)
                ctx.decode_batch_ex(1, ph, None, **outs)
                ctx.synchronize()
            for w in ("z", "U"):
                assert np.array_equal(outs[w][0].cpu().numpy(), ref[w], equal_nan=True), (T, variant, w)


@pytest.mark.parametrize("name", ["C1x4", "C2", "C3", "C5"])
def test_every_item_length_and_launch_size(api, oracle, synth, torch_cuda, name):
    """The launcher picks the rows per work item from the launch's size (1, 2, 3, 5, 6, 10, 12, 16 ... rows, with or without the
    tiered tail): every length 1..16 forced on a ragged batch, and the automatic choice for batches of 1 to 9 frame-sets, in
    every strip-kernel mode, against the oracle."""
    torch = torch_cuda
    spec = small_spec(synth, name, 516, 71)
    n_max = 9
    sets = [synth.random_planes(spec, seed=4000 + s) for s in range(n_max)]
    refs = [oracle.pipeline(spec, p, g, want=("z",))["z"] for p, g in sets]
    ph = torch.from_numpy(np.stack([p for p, _ in sets])).cuda()
    gr = None if sets[0][1] is None else torch.from_numpy(np.stack([g for _, g in sets])).cuda()
    H, W = spec["height"], spec["width"]
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        ctx.set_variant(2)
        for rows, n in [(r, 3) for r in range(1, 17)] + [(0, n) for n in range(1, n_max + 1)]:
            ctx.set_tuning(strip_rows=rows)
            z = torch.full((n, H, W), -5.0, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            ctx.decode_batch(n, ph[:n], None if gr is None else gr[:n], z)
            ctx.synchronize()
            got = z.cpu().numpy()
            for s in range(n):
                assert np.array_equal(got[s], refs[s], equal_nan=True), (rows, n, s)


@pytest.mark.parametrize("rows,tail_pct,tail_rows,tiers", [(8, 20, 2, 2), (16, 30, 4, 3), (8, 50, 1, 4), (12, 10, 3, 3), (16, 60, 4, 4), (32, 70, 8, 4)])
@pytest.mark.parametrize("shape", [(67, 256), (130, 1000), (200, 64), (97, 1920)])
def test_strip_kernel_long_and_short_items(api, oracle, synth, shape, rows, tail_pct, tail_rows, tiers):
    """The tiered item layout (long items first, the last rows of every frame-set in ever shorter items) is chosen
    automatically only for large launches; slx_set_tuning forces it here on small, ragged tiles, for a batch of 3
    frame-sets, in every strip-kernel mode (the Gray-mask mode re-groups each tier's workgroups by XCD)."""
    import torch
    h, w = shape
    for name in ("C1x4", "C4", "C5", "C3"):
        spec = small_spec(synth, name, w, h)
        sets = [synth.random_planes(spec, seed=7 * h + w + i) for i in range(3)]
        want = [oracle.pipeline(spec, ph, gr, want=("z",))["z"] for ph, gr in sets]
        phase = torch.from_numpy(np.stack([ph for ph, _ in sets])).cuda()
        gray = torch.from_numpy(np.stack([gr for _, gr in sets])).cuda() if sets[0][1] is not None else None
        z = torch.full((3, h, w), -1.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        with api.Context(spec) as ctx:
            ctx.set_variant(2)
            ctx.set_tuning(strip_rows=rows, tail_pct=tail_pct, tail_rows=tail_rows, tiers=tiers)
            ctx.decode_batch(3, phase, gray, z)
            ctx.synchronize()
        got = z.cpu().numpy()
        for i in range(3):
            assert np.array_equal(got[i], want[i], equal_nan=True), (name, i)


def test_environment_is_ignored(api, oracle, synth, monkeypatch):
    """The library never reads the process environment: the round-1 debug variables (SLX_DBG=1 used to drop every
    depth store) change nothing, and out-of-range tuning values are refused."""
    for k, v in (("SLX_DBG", "1"), ("SLX_STRIP_ROWS", "3"), ("SLX_TAIL_PCT", "50"), ("SLX_TAIL_ROWS", "1"), ("SLX_GRAY_PLAIN", "1"),
                 ("SLX_STRIP_WAVES", "1"), ("SLX_LDS_PAD", "64")):
        monkeypatch.setenv(k, v)
    for name in ("C3", "C4"):
        spec = small_spec(synth, name, 256, 70)
        ph, gr = synth.random_planes(spec, seed=31)
        ref = oracle.pipeline(spec, ph, gr, want=("z",))
        got = api.decode_frameset(spec, ph, gr, want=("z",), variant=api.VARIANT_STRIP)
        assert_same(got, ref, ("z",))
    with api.Context(small_spec(synth, "C4", 64, 8)) as ctx:
        for bad in (dict(strip_rows=33), dict(strip_rows=-1), dict(tail_pct=100), dict(strip_waves=5), dict(lds_pad_kib=129), dict(tiers=5)):
            with pytest.raises(api.SlxError) as e:
                ctx.set_tuning(**bad)
            assert e.value.code == api.ERR_INVALID_ARG


@pytest.mark.parametrize("tune", [dict(gray_plain=1), dict(strip_waves=2), dict(plain_order=1), dict(lds_pad_kib=40), dict(strip_rows=1), dict(tail_pct=-1), dict(tiers=1),
                                  dict(strip_rows=8, tiers=4, tail_pct=60, tail_rows=2)])
def test_tuning_keys_do_not_change_results(api, oracle, synth, tune):
    spec = small_spec(synth, "C3", 500, 67)
    ph, gr = synth.random_planes(spec, seed=17)
    gr[:, :, :250] = np.where(gr[:, :, :250] > 127, 220, 20)
    ref = oracle.pipeline(spec, ph, gr, want=("z",))["z"]
    with api.Context(spec) as ctx:
        ctx.set_variant(api.VARIANT_STRIP)
        ctx.set_tuning(**tune)
        ctx.set_frames(ph, gr)
        ctx.decode()
        assert np.array_equal(ctx.get_depth(), ref, equal_nan=True)


def test_strip_variant_refuses_ineligible_operands(api, synth):
    spec = small_spec(synth, "C2", 63, 8)                   # width not a multiple of 4
    ph, _ = synth.random_planes(spec, seed=1)
    with api.Context(spec) as ctx:
        ctx.set_variant(api.VARIANT_STRIP)
        ctx.set_frames(ph, None)
        with pytest.raises(api.SlxError) as e:
            ctx.decode()
        assert e.value.code == api.ERR_UNAVAILABLE
        ctx.set_variant(api.VARIANT_AUTO)                   # automatic selection falls back to the generic kernel
        ctx.decode()
    big = dict(small_spec(synth, "C2", 64, 8), periods=[40000, 160, 20])   # period > 2^14: exact-division kernel
    ph, _ = synth.random_planes(big, seed=2)
    with api.Context(big) as ctx:
        ctx.set_variant(api.VARIANT_STRIP)
        ctx.set_frames(ph, None)
        with pytest.raises(api.SlxError):
            ctx.decode()


def test_degenerate_depth_quotients(api, oracle, synth):
    """den == 0, num == 0 and 0/0 in z = -(cA - cB U)/(cC - cD U): the in-range fast division must hand
    these to the IEEE division (inf -> FOV clamp -> 0, NaN stays NaN as in the reference's compare chain)."""
    spec = small_spec(synth, "C1", 64, 8)
    cal = {"cam": [1.0, 0, 0.0, 0, 1.0, 0.0, 0, 0, 1], "pro": [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0],
           "rot": [1.0, 0, 0, 0, 1.0, 0, 0, 0, 0.0], "trans": [0.0, 0.0, 0.0]}      # P row 2 == 0, cA == cB == 0
    spec["calib"] = cal
    ph, _ = synth.random_planes(spec, seed=5)
    ref = oracle.pipeline(spec, ph, None, want=("z",))
    for variant in (0, 1, 2, 3):
        got = api.decode_frameset(spec, ph, None, want=("z",), variant=variant)
        assert np.array_equal(got["z"], ref["z"], equal_nan=True), variant
    cal2 = dict(cal, rot=[1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0], trans=[0.0, 0.0, 0.0])  # num == 0 everywhere, den varies
    spec["calib"] = cal2
    ref = oracle.pipeline(spec, ph, None, want=("z",))
    for variant in (0, 1, 2, 3):
        got = api.decode_frameset(spec, ph, None, want=("z",), variant=variant)
        assert np.array_equal(got["z"], ref["z"], equal_nan=True), variant


# ------------------------------------------------------------------ the boundary itself
def test_device_frames_strides_and_batch(api, oracle, synth, torch_cuda):
    torch = torch_cuda
    spec = small_spec(synth, "C3", 200, 40)
    n_sets = 5
    sets = [synth.random_planes(spec, seed=50 + s) for s in range(n_sets)]
    ref = [oracle.pipeline(spec, p, g, want=("z",))["z"] for p, g in sets]
    H, W = spec["height"], spec["width"]
    for pitch in (W, W + 8, W + 3):                      # dense, padded-aligned, padded-unaligned rows
        ph = torch.zeros((n_sets, 12, H, pitch), dtype=torch.uint8, device="cuda")
        gr = torch.zeros((n_sets, 12, H, pitch), dtype=torch.uint8, device="cuda")
        for s, (p, g) in enumerate(sets):
            ph[s, :, :, :W] = torch.from_numpy(p).cuda()
            gr[s, :, :, :W] = torch.from_numpy(g).cuda()
        torch.cuda.synchronize()              # the context's stream does not order itself against torch's stream
        with api.Context(spec) as ctx:
            # borrowed device planes, one frame-set at a time
            for s in range(n_sets):
                ctx.set_frames(ph[s, :, :, :W], gr[s, :, :, :W])
                ctx.decode()
                assert np.array_equal(ctx.get_depth(), ref[s], equal_nan=True), (pitch, s)
            # one launch for the whole batch
            z = torch.full((n_sets, H, W), -1.0, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()          # the context's stream does not order itself against torch's stream
            ctx.decode_batch(n_sets, ph, gr, z, row_stride=pitch)
            ctx.synchronize()
            torch.cuda.synchronize()
            zb = z.cpu().numpy()
            for s in range(n_sets):
                assert np.array_equal(zb[s], ref[s], equal_nan=True), (pitch, s)


def test_frames_of_different_row_strides_in_one_decode(api, oracle, synth, torch_cuda):
    """SetMat takes any image (pic.copyTo, R/CDecodePhase.cpp:114): host frames (staged at the width rounded up to 4), strided host
    frames and device frames borrowed from buffers of three different pitches, mixed within ONE frame-set and changed between
    decodes -- on the context's stream and on a caller's.  (tools/fuzz_api.py met the error this used to be.)"""
    torch = torch_cuda
    side = torch.cuda.Stream()
    for name, w, h in (("C3", 124, 64), ("C1x4", 156, 9), ("C2", 78, 33), ("C3", 61, 40)):
        spec = small_spec(synth, name, w, h)
        ph, gr = synth.random_planes(spec, seed=w + h)
        with api.Context(spec) as ctx:
            keep = []
            for rnd in range(3):
                for grp, planes in ((api.GROUP_PHASE, ph), (api.GROUP_GRAY, gr)):
                    for i in range(0 if planes is None else planes.shape[0]):
                        planes[i] = np.random.default_rng(1000 * rnd + 37 * i + grp).integers(0, 256, size=(h, w), dtype=np.uint8)
                        how = (i + rnd + grp) % 4
                        if how == 0:
                            ctx.set_frame(grp, i, planes[i])
                        elif how == 1:
                            wide = np.zeros((h, w + 12), dtype=np.uint8)
                            wide[:, :w] = planes[i]
                            ctx.set_frame(grp, i, wide[:, :w])
                        else:
                            dev = torch.zeros((h, w + (0, 0, 4, 64)[how] + 8 * rnd), dtype=torch.uint8, device="cuda")
                            dev[:, :w] = torch.from_numpy(planes[i]).cuda()
                            keep.append(dev)
                            ctx.set_frame(grp, i, dev[:, :w])
                torch.cuda.synchronize()
                ctx.decode(stream=side.cuda_stream if rnd == 1 else None)
                ref = oracle.pipeline(spec, ph, gr, want=("z",))["z"]
                assert np.array_equal(ctx.get_depth(), ref, equal_nan=True), (name, w, h, rnd)


def test_decode_on_caller_stream_and_timing(api, oracle, synth, torch_cuda):
    torch = torch_cuda
    spec = small_spec(synth, "C2", 256, 64)
    ph, _ = synth.random_planes(spec, seed=9)
    ref = oracle.pipeline(spec, ph, None, want=("z",))["z"]
    dev = torch.from_numpy(ph).cuda()
    z = torch.empty((1, 64, 256), dtype=torch.float64, device="cuda")
    s = torch.cuda.Stream()
    with api.Context(spec) as ctx:
        ctx.enable_timing(True)
        with torch.cuda.stream(s):
            ctx.decode_batch(1, dev[None], None, z, stream=s.cuda_stream)
        ms = ctx.last_decode_ms()
        s.synchronize()
        assert ms > 0.0
        assert np.array_equal(z[0].cpu().numpy(), ref, equal_nan=True)
        # idempotence: a second decode of the same inputs changes nothing
        ctx.decode_batch(1, dev[None], None, z, stream=s.cuda_stream)
        s.synchronize()
        assert np.array_equal(z[0].cpu().numpy(), ref, equal_nan=True)


def _hip_runtime():
    """The HIP runtime already loaded into this process, for a stream the test itself creates and destroys (a torch stream comes
    from torch's pool and is never destroyed)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    return hip, ctypes


def test_caller_stream_destroyed_right_after_the_decode(api, oracle, synth, torch_cuda):
    """Round-2 fault (a lazily recorded event on a stream the caller had destroyed crashed slx_destroy): the library keeps no
    handle of a caller's stream.  Decode on a stream of our own, destroy the stream at once, then read, decode on a NEW stream
    (which may get the old handle value), read again, close."""
    torch = torch_cuda
    hip, ctypes = _hip_runtime()
    spec = small_spec(synth, "C2", 256, 64)
    sets = [synth.random_planes(spec, seed=70 + i)[0] for i in range(3)]
    refs = [oracle.pipeline(spec, p, None, want=("z",))["z"] for p in sets]
    devs = [torch.from_numpy(p).cuda() for p in sets]
    torch.cuda.synchronize()
    ctx = api.Context(spec)
    for i in range(3):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0          # hipStreamNonBlocking
        ctx.set_frames(devs[i], None)
        ctx.decode(stream=s.value)
        assert hip.hipStreamDestroy(s) == 0                                    # pending work completes, the handle is dead
        assert np.array_equal(ctx.get_depth(), refs[i], equal_nan=True), i
    # the last launch ran on a stream that no longer exists: the own-stream launch that follows is ordered through the event
    ctx.set_frames(devs[0], None)
    ctx.decode()
    assert np.array_equal(ctx.get_depth(), refs[0], equal_nan=True)
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    ctx.set_frames(devs[1], None)
    ctx.decode(stream=s.value)
    assert hip.hipStreamDestroy(s) == 0
    ctx.close()                                                                # waits through the event, then frees


def test_pipe_destroyed_before_its_context(api, oracle, synth):
    """The pipe's decode stream is a caller's stream to the context: destroying the pipe (and its streams) first, then reading
    from and destroying the context, must neither crash nor lose the ordering."""
    spec = small_spec(synth, "C2", 128, 40)
    ph, _ = synth.random_planes(spec, seed=77)
    ref = oracle.pipeline(spec, ph, None, want=("z",))["z"]
    ctx = api.Context(spec)
    pipe = api.Pipe(ctx, slots=2, sets_per_slot=1, host_result=True)
    W = spec["width"]
    buf = pipe.acquire()
    buf[0, :, :, :W] = ph
    pipe.submit(1)
    got = np.array(pipe.collect()[0], copy=True)
    buf = pipe.acquire()
    buf[0, :, :, :W] = ph
    pipe.submit(1)                      # still in flight when the pipe goes
    pipe.close()
    ctx.set_frames(ph, None)
    ctx.decode()
    assert np.array_equal(ctx.get_depth(), ref, equal_nan=True)
    ctx.close()
    assert np.array_equal(got.reshape(ref.shape), ref, equal_nan=True)


def test_context_destroyed_before_its_pipe(api, oracle, synth):
    """The other order: the context goes first, with a slot of the pipe still in flight; the pipe is destroyed afterwards (it
    keeps the device number itself and waits for its own streams) and a second context on the same device works on."""
    spec = small_spec(synth, "C2", 128, 40)
    ph, _ = synth.random_planes(spec, seed=78)
    ref = oracle.pipeline(spec, ph, None, want=("z",))["z"]
    ctx = api.Context(spec)
    pipe = api.Pipe(ctx, slots=3, sets_per_slot=2, host_result=True)
    W = spec["width"]
    for _ in range(2):
        buf = pipe.acquire()
        buf[0, :, :, :W] = ph
        buf[1, :, :, :W] = ph
        pipe.submit(2)
    got = np.array(pipe.collect(), copy=True)
    ctx.close()                         # one slot is still submitted
    pipe.close()
    assert np.array_equal(got[0], ref, equal_nan=True) and np.array_equal(got[1], ref, equal_nan=True)
    with api.Context(spec) as again:
        again.set_frames(ph, None)
        again.decode()
        assert np.array_equal(again.get_depth(), ref, equal_nan=True)


def test_contexts_on_concurrent_host_threads(api, oracle, synth, torch_cuda):
    """One context per host thread (the library keeps no state outside a context but a thread-local error string; ctypes drops
    the GIL during a call, so the calls really overlap): four threads, four configurations, host-fed single decodes, batch
    decodes on a stream of their own, read-backs and point clouds interleaved, every result against the oracle."""
    import threading
    torch = torch_cuda
    jobs = []
    for k, (name, w, h) in enumerate((("C2", 256, 48), ("C3", 200, 40), ("C1x4", 128, 64), ("C5", 192, 33))):
        spec = small_spec(synth, name, w, h)
        sets = [synth.random_planes(spec, seed=900 + 10 * k + s) for s in range(3)]
        refs = [oracle.pipeline(spec, p, g, want=("z",))["z"] for p, g in sets]
        jobs.append((spec, sets, refs))
    errors = []

    def worker(spec, sets, refs):
        try:
            H, W = spec["height"], spec["width"]
            stream = torch.cuda.Stream()
            dev_p = torch.from_numpy(np.stack([p for p, _ in sets])).cuda()
            dev_g = None if sets[0][1] is None else torch.from_numpy(np.stack([g for _, g in sets])).cuda()
            torch.cuda.synchronize()
            with api.Context(spec) as ctx:
                for rep in range(12):
                    s = rep % len(sets)
                    ctx.set_frames(*sets[s])
                    ctx.decode()
                    assert np.array_equal(ctx.get_depth(), refs[s], equal_nan=True), ("decode", spec["name"], rep)
                    if rep % 3 == 0:
                        assert np.array_equal(ctx.get_point_cloud(), oracle.point_cloud(spec, refs[s])), ("cloud", spec["name"], rep)
                    z = torch.full((len(sets), H, W), -1.0, dtype=torch.float64, device="cuda")
                    torch.cuda.synchronize()
                    ctx.decode_batch(len(sets), dev_p, dev_g, z, stream=stream.cuda_stream)
                    stream.synchronize()
                    for q in range(len(sets)):
                        assert np.array_equal(z[q].cpu().numpy(), refs[q], equal_nan=True), ("batch", spec["name"], rep, q)
        except BaseException as e:                # reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)


def test_fused_point_cloud_beside_a_chip_filling_decode(api, oracle, synth, torch_cuda):
    """The fused point-cloud kernel's look-back spins on words other workgroups publish; its liveness argument (slx_cloud.hip) must
    hold when the chip is full of somebody else's waves.  Thread A keeps a second context decoding configuration 4's 32-frame-set
    batch back to back (a launch that fills every CU many times over, the stream kernel's resident waves); thread B takes 150
    clouds of two 1920 x 1200 depth maps through the fused launch meanwhile.  Every cloud must be the oracle's; both threads finish."""
    import threading
    torch = torch_cuda
    spec = synth.make_spec("C4")
    H, W = spec["height"], spec["width"]
    rng = np.random.default_rng(4242)
    planes = [rng.uniform(50.0, 1200.0, size=(H, W)) for _ in range(2)]
    planes[1][:, ::7] = 5000.0
    refs = [oracle.point_cloud(spec, zz) for zz in planes]
    z = torch.from_numpy(np.stack(planes)).cuda()
    batch = torch.randint(0, 256, (32, 12, H, W), dtype=torch.uint8, device="cuda")
    zb = torch.empty((32, H, W), dtype=torch.float64, device="cuda")
    dev = torch.empty((H * W, 3), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    stop, errors, done = threading.Event(), [], {"decodes": 0, "clouds": 0}

    def decoder():
        try:
            with api.Context(spec) as ctx:
                while not stop.is_set():
                    for _ in range(20):
                        ctx.decode_batch(32, batch, None, zb)
                    ctx.synchronize()
                    done["decodes"] += 20
        except BaseException as e:
            errors.append("decoder: %r" % (e,))

    def clouds():
        try:
            with api.Context(spec) as ctx:
                ctx.set_tuning(cloud_passes=1)
                for rep in range(150):
                    k = rep & 1
                    n = ctx.point_cloud_of_depth(z[k], out=dev)
                    assert n == len(refs[k]), (rep, n, len(refs[k]))
                    if rep % 10 == 0:
                        assert np.array_equal(dev[:n].cpu().numpy(), refs[k]), rep
                    done["clouds"] += 1
        except BaseException as e:
            errors.append("clouds: %r" % (e,))
        finally:
            stop.set()

    ta, tb = threading.Thread(target=decoder), threading.Thread(target=clouds)
    ta.start()
    tb.start()
    tb.join(240)
    stop.set()
    ta.join(60)
    assert not errors, errors
    assert not ta.is_alive() and not tb.is_alive()
    assert done["clouds"] == 150 and done["decodes"] >= 20, done


def test_error_paths_on_device(api, synth):
    spec = small_spec(synth, "C1x4", 32, 8)
    ph, gr = synth.random_planes(spec, seed=1)
    with api.Context(spec) as ctx:
        with pytest.raises(api.SlxError) as e:
            ctx.get_depth()
        assert e.value.code == api.ERR_NOT_DECODED
        with pytest.raises(api.SlxError) as e:
            ctx.decode()                                   # nothing set yet
        assert e.value.code == api.ERR_MISSING_FRAME
        with pytest.raises(api.SlxError) as e:
            ctx.set_frame(api.GROUP_PHASE, 4, ph[0])       # only 4 phase images exist (R/CDecodePhase.cpp:109)
        assert e.value.code == api.ERR_NOT_CONFIGURED
        with pytest.raises(api.SlxError) as e:
            ctx.get_output("k")
        assert e.value.code == api.ERR_UNAVAILABLE
        ctx.set_frames(ph, gr)
        ctx.decode()
        assert ctx.get_depth().shape == (8, 32)
    with api.Context(dict(spec, mode=synth.MODE_PHASE_ONLY)) as ctx:
        with pytest.raises(api.SlxError) as e:
            ctx.set_frame(api.GROUP_GRAY, 0, gr[0])        # this mode has no Gray planes
        assert e.value.code == api.ERR_NOT_CONFIGURED


def test_calibration_constants(api, oracle, synth):
    spec = synth.make_spec("C4")
    with api.Context(spec) as ctx:
        P, cA, cB = ctx.get_calibration()
    Po = oracle.projection_matrix(spec["calib"]["pro"], spec["calib"]["rot"], spec["calib"]["trans"])
    assert np.array_equal(P, Po)
    fu, fv = spec["calib"]["cam"][0], spec["calib"]["cam"][4]
    assert cA == fu * fv * Po[0, 3] and cB == fu * fv * Po[2, 3]


def test_gray_lut_replacement(api, oracle, synth):
    spec = small_spec(synth, "C1x4", 64, 16)
    ph, gr = synth.random_planes(spec, seed=4)
    lut2 = (np.arange(64)[::-1] - 10).astype(np.int16)      # an arbitrary table, negative entries included
    ref = oracle.pipeline(dict(spec, gray_lut=lut2), ph, gr, want=("z", "U", "gray"))
    with api.Context(spec, aux=("U", "gray")) as ctx:
        ctx.set_gray_lut(lut2)
        ctx.set_frames(ph, gr)
        ctx.decode()
        for w in ("z", "U", "gray"):
            assert np.array_equal(ctx.get_output(w), ref[w], equal_nan=True), w


# ------------------------------------------------------------------ point cloud (CCalculation::Result)
@pytest.mark.parametrize("name,shape", [("C1x4", (120, 200)), ("C4", (1200, 1920)), ("C3", (33, 130)), ("C2", (1, 4))])
@pytest.mark.parametrize("passes", [0, 2])
def test_point_cloud(api, oracle, synth, name, shape, passes):
    """CCalculation::Result's data (R/CCalculation.cpp:323-357, :756-771) against the oracle, byte for byte: passes = 0 the
    library's choice -- the single fused launch that reads the depth once (slx_cloud.hip) on every shape here -- and 2 the count +
    write launches of rounds 1-4."""
    h, w = shape
    spec = small_spec(synth, name, w, h) if (h, w) != (1200, 1920) else synth.make_spec(name)
    ph, gr, _ = synth.render(spec, "sphere", seed=9, noise_sigma=2.0)
    z = oracle.pipeline(spec, ph, gr, want=("z",), threads=8)["z"]
    ref = oracle.point_cloud(spec, z)
    with api.Context(spec) as ctx:
        ctx.set_tuning(cloud_passes=passes)
        ctx.set_frames(ph, gr)
        ctx.decode()
        got = ctx.get_point_cloud()
        view = ctx.get_point_cloud_view()                  # the same cloud in the context's pinned memory, twice (the buffer is reused)
        assert not view.flags.writeable and np.array_equal(view, ref)
        assert np.array_equal(np.array(ctx.get_point_cloud_view()), ref)
    assert got.shape == ref.shape and ref.shape[0] > 0
    assert np.array_equal(got, ref)
    # nothing in the FOV -> an empty cloud, not an error
    with api.Context(dict(spec, fov_min=1e9, fov_max=2e9)) as ctx:
        with pytest.raises(api.SlxError):
            ctx.get_point_cloud_view()                     # before any decode
        ctx.set_frames(ph, gr)
        ctx.decode()
        assert ctx.get_point_cloud().shape == (0, 3) and ctx.get_point_cloud_view().shape == (0, 3)


@pytest.mark.parametrize("shape", [(1, 1), (3, 17), (33, 16), (255, 31), (256, 48), (257, 33), (700, 130), (1200, 1920), (3000, 250), (4500, 40), (5000, 20),
                                   (3000, 4096)])
def test_fused_point_cloud_geometries(api, oracle, synth, torch_cuda, shape):
    """slx_cloud_fused_kernel's work split -- column groups of 16 (ragged last group), parts of 256 rows (ragged last part, one part
    shorter than a pass of the lanes, more rows per part on maps taller than 16 x 256) -- on depth planes of unstructured content
    (a third of the depths outside the FOV, NaNs and infinities among them) through slx_point_cloud_of_depth: cloud_passes = 1
    (the fused launch or an error) must equal the two-launch path and the oracle's order byte for byte; launch after launch on
    one context (the ticket counters and the epoch tags carry over), into host memory, into a device buffer, count only.  The last
    shape (config 5's 4096 x 3000) is 3 072 workgroups for the 768 the chip keeps resident: the look-back's waits must also
    resolve when the parts run in several rounds."""
    torch = torch_cuda
    h, w = shape
    spec = small_spec(synth, "C4", w, h)
    spec["fov_min"], spec["fov_max"] = 100.0, 900.0
    rng = np.random.default_rng(h * 131 + w)
    planes = []
    for k in range(3):
        zz = rng.uniform(50.0, 1200.0, size=(h, w))
        zz[rng.random((h, w)) < 0.02] = np.nan
        zz[rng.random((h, w)) < 0.02] = np.inf
        zz[rng.random((h, w)) < 0.05] = 0.0
        if k == 2:
            zz[:, : w // 2] = 2000.0                                  # whole columns without a point
        planes.append(zz)
    z = torch.from_numpy(np.stack(planes)).cuda()
    dev = torch.full((h * w, 3), -7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        for k in (0, 1, 2, 0):
            ref = oracle.point_cloud(spec, planes[k])
            ctx.set_tuning(cloud_passes=1)
            got = ctx.point_cloud_of_depth(z[k])
            assert got.shape == ref.shape and np.array_equal(got, ref, equal_nan=True), (k, "fused, host")
            n = ctx.point_cloud_of_depth(z[k], out=dev)
            assert n == len(ref) and np.array_equal(dev[:n].cpu().numpy(), ref, equal_nan=True), (k, "fused, device")
            ctx.set_tuning(cloud_passes=2)
            assert np.array_equal(ctx.point_cloud_of_depth(z[k]), ref, equal_nan=True), (k, "two launches")
        # a device buffer smaller than the frame: the count comes first, then the points (two fused launches)
        ctx.set_tuning(cloud_passes=1)
        ref = oracle.point_cloud(spec, planes[1])
        if 0 < len(ref) < h * w:
            tight = torch.full((len(ref), 3), -7.0, dtype=torch.float64, device="cuda")
            assert ctx.point_cloud_of_depth(z[1], out=tight) == len(ref) and np.array_equal(tight.cpu().numpy(), ref, equal_nan=True)


@pytest.mark.parametrize("shape", [(700, 130), (1200, 1920)])
def test_fused_point_cloud_gives_up_instead_of_hanging(api, oracle, synth, torch_cuda, shape):
    """The look-back of slx_cloud_fused_kernel polls words other workgroups publish; its progress rests on how the dispatcher hands out
    workgroups.  The spin is BOUNDED: a workgroup whose words do not arrive within the bound raises a flag and writes nothing, and the
    host repeats the frame on the count + write launches.  Forced here through SLX_TUNE_CLOUD_SPIN = 1 (no poll at all: every workgroup
    that needs a word of another one gives up): the cloud must still equal the oracle's, the call must still succeed, slx_last_error
    says what happened, and the next frame with the default bound takes the fused launch again (the tagged words were re-zeroed)."""
    torch = torch_cuda
    h, w = shape
    spec = small_spec(synth, "C4", w, h)
    spec["fov_min"], spec["fov_max"] = 100.0, 900.0
    rng = np.random.default_rng(h * 7 + w)
    planes = [rng.uniform(50.0, 1200.0, size=(h, w)) for _ in range(2)]
    z = torch.from_numpy(np.stack(planes)).cuda()
    dev = torch.full((h * w, 3), -7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        refs = [oracle.point_cloud(spec, pl) for pl in planes]
        ctx.set_tuning(cloud_passes=1)
        assert np.array_equal(ctx.point_cloud_of_depth(z[0]), refs[0])                  # the fused launch, default bound
        assert "gave up" not in ctx.last_error()
        ctx.set_tuning(cloud_passes=1, cloud_spin=1)
        for k in (1, 0, 1):
            got = ctx.point_cloud_of_depth(z[k])
            assert got.shape == refs[k].shape and np.array_equal(got, refs[k]), (k, "host")
            n = ctx.point_cloud_of_depth(z[k], out=dev)
            assert n == len(refs[k]) and np.array_equal(dev[:n].cpu().numpy(), refs[k]), (k, "device")
        assert "gave up" in ctx.last_error()                                             # (at least one of those frames fell back)
        ctx.set_tuning(cloud_passes=1, cloud_spin=0)
        for k in (0, 1):
            assert np.array_equal(ctx.point_cloud_of_depth(z[k]), refs[k]), (k, "after the fallbacks")


@pytest.mark.parametrize("shape", [(300, 70), (4500, 40)])
def test_fused_point_cloud_with_focal_lengths_outside_the_cheap_division(api, oracle, synth, torch_cuda, shape):
    """x = z (u - cx) / fu, y = z (v - cy) / fv (R/CCalculation.cpp:756-771) by the literal f64 division: the instantiations of
    slx_cloud_fused_kernel for a focal length outside the range of the refined-reciprocal sequence (|f| >= 2^90 here) -- never a real
    calibration, but the kernel has the code and it must be the oracle's.  Parts of 256 rows and the tall parts of a 4500-row map."""
    torch = torch_cuda
    h, w = shape
    spec = small_spec(synth, "C4", w, h)
    spec["fov_min"], spec["fov_max"] = 100.0, 900.0
    cal = dict(spec["calib"])
    cam = list(cal["cam"])
    cam[0], cam[4] = 2.0 ** 95, -(2.0 ** 93)
    cal["cam"] = cam
    spec["calib"] = cal
    rng = np.random.default_rng(h + w)
    plane = rng.uniform(50.0, 1200.0, size=(h, w))
    ref = oracle.point_cloud(spec, plane)
    z = torch.from_numpy(plane).cuda()
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        for passes in (1, 2, 1):
            ctx.set_tuning(cloud_passes=passes)
            got = ctx.point_cloud_of_depth(z)
            assert got.shape == ref.shape and len(ref) > 0 and np.array_equal(got, ref), passes
    assert np.all(np.abs(ref[:, 0]) < 1e-20)                                             # (the division really ran against 2^95)


# ------------------------------------------------------------------ dynamic frames (CCalculation::CalculateOther)
def dyna_images(h, w, n, seed):
    """A moving stripe pattern seen by the camera, with noise: what the tracker's column-sum extrema follow."""
    rng = np.random.default_rng(seed)
    u = np.arange(w)[None, :] + 0.03 * np.arange(h)[:, None]
    out = []
    for f in range(n):
        img = 128 + 100 * np.sign(np.sin(2 * np.pi * (u + 1.7 * f) / 14.0)) + rng.normal(0, 6, (h, w))
        out.append(np.clip(img, 0, 255).astype(np.uint8))
    return out


@pytest.mark.parametrize("shape,window", [((96, 160), 21), ((40, 300), 21), ((25, 23), 21), ((64, 64), 5), ((20, 64), 21),
                                          ((130, 256), 21), ((30, 257), 21), ((70, 492), 21), ((24, 1000), 9),
                                          # band / tile edges of the 21-pixel kernel: a single interior pixel, one interior row,
                                          # bands of 8 rows starting exactly at / next to the interior, exactly one and two tiles
                                          ((21, 21), 21), ((21, 80), 21), ((22, 45), 21), ((29, 64), 21), ((37, 256), 21), ((34, 472), 21),
                                          ((27, 473), 21),
                                          # 254 output columns per workgroup (round 5): exactly one tile, one column more, the halo lane on the
                                          # last image column, two tiles and one more, odd widths (the sums go two columns per lane)
                                          ((26, 254), 21), ((26, 255), 21), ((23, 253), 21), ((30, 508), 21), ((26, 509), 21), ((31, 763), 21),
                                          ((26, 240), 21), ((41, 433), 21), ((72, 305), 21), ((130, 192), 21)])
def test_dynamic_frames(api, oracle, synth, shape, window):
    h, w = shape
    spec = small_spec(synth, "C1x4", w, h)
    ph, gr, _ = synth.render(spec, "tilted", noise_sigma=2.0)
    ref0 = oracle.pipeline(spec, ph, gr, want=("z", "U"))
    imgs = dyna_images(h, w, 5, seed=h + w)
    with api.Context(spec, aux=("U", "x", "y")) as ctx:
        with pytest.raises(api.SlxError) as e:
            ctx.track_next(imgs[1])
        assert e.value.code == api.ERR_NOT_CONFIGURED
        ctx.set_frames(ph, gr)
        ctx.decode()
        with pytest.raises(api.SlxError):
            ctx.track_begin(imgs[0], window=20)               # even window
        ctx.track_begin(imgs[0], window=window)
        sw0, sb0 = oracle.strip_regression(imgs[0], window)
        assert np.array_equal(ctx.get_output("stripW"), sw0) and np.array_equal(ctx.get_output("stripB"), sb0)
        U, z_prev = ref0["U"], ref0["z"]
        for f in range(1, 5):
            ctx.track_next(imgs[f])
            sw1, sb1 = oracle.strip_regression(imgs[f], window)
            dP = oracle.delta_p(sw0, sb0, sw1, sb1)
            U = U + dP.astype(np.float64)
            tri = oracle.triangulate(spec, U)
            for name, want in (("stripW", sw1), ("stripB", sb1), ("deltaP", dP), ("U", U), ("z", tri["z"]), ("x", tri["x"]),
                               ("y", tri["y"]), ("deltaZ", tri["z"] - z_prev)):
                assert np.array_equal(ctx.get_output(name), want, equal_nan=True), (f, name)
            assert np.array_equal(ctx.get_point_cloud(), oracle.point_cloud(spec, tri["z"]))
            sw0, sb0, z_prev = sw1, sb1, tri["z"]
        if h > 2 * window and w > 2 * window:
            assert np.any(sw0 != 0) and np.any(dP != 0)
    with api.Context(spec) as ctx:                            # no U plane: the tracker refuses
        ctx.set_frames(ph, gr)
        ctx.decode()
        with pytest.raises(api.SlxError) as e:
            ctx.track_begin(imgs[0])
        assert e.value.code == api.ERR_UNAVAILABLE


@pytest.mark.filterwarnings("ignore:The CUDA Graph is empty")       # (the capture this test expects to be refused)
def test_decodes_captured_into_a_hip_graph(api, oracle, synth, torch_cuda):
    """slx_decode_batch_ex on a caller's stream inside a stream capture (include/slx.h, at slx_decode): four single-frame-set launches and a
    32-set batch -- on the STREAM kernel, whose queue counters return to zero at the end of every launch (round 6; before, a captured
    batch had to fall back to the strip kernel) -- become one graph; three replays over changing inputs, with plain launches of the
    same context in between, give what plain launches give, frame-set 0 also against the oracle; a capture that begins while work of
    the context is in flight is refused, not mis-ordered."""
    torch = torch_cuda
    spec = small_spec(synth, "C4", 1920, 304)
    H, W = spec["height"], spec["width"]
    ph = torch.randint(0, 256, (36, 12, H, W), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(21))
    z = torch.empty((36, H, W), dtype=torch.float64, device="cuda")
    want = torch.empty_like(z)
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        def launches(out, stream=None):
            for r in range(4):
                ctx.decode_batch_ex(1, ph[r:r + 1], None, z=out[r:r + 1], stream=stream)
            ctx.decode_batch_ex(32, ph[4:], None, z=out[4:], stream=stream)
        ctx.set_tuning(stream=2)                                              # the stream kernel whenever it can run ...
        launches(want)
        assert ctx.last_kernel().startswith("slx_stream_kernel<3>")          # ... which a plain 32-set launch can
        ctx.synchronize()
        ref = oracle.pipeline(spec, ph[0].cpu().numpy(), None, want=("z",))["z"]
        assert np.array_equal(want[0].cpu().numpy(), ref, equal_nan=True)
        s = torch.cuda.Stream()
        ctx.decode_batch_ex(1, ph[0:1], None, z=z[0:1])                       # in flight on the context's stream ...
        g = torch.cuda.CUDAGraph()
        with pytest.raises(api.SlxError) as e:                                # ... so a capture may not begin
            with torch.cuda.graph(g, stream=s):
                ctx.decode_batch_ex(1, ph[0:1], None, z=z[0:1], stream=s.cuda_stream)
        assert e.value.code == api.ERR_INVALID_ARG
        ctx.synchronize()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            launches(z, stream=s.cuda_stream)
        assert ctx.last_kernel().startswith("slx_stream_kernel<3>"), ctx.last_kernel()   # the captured batch is a stream-kernel launch
        for rep in range(3):
            z.fill_(-1.0)
            if rep:
                ph.copy_(ph.flip(0))                                          # the graph reads the buffers as they are at replay
                torch.cuda.synchronize()                                      # (torch's stream: the context's own does not wait for it)
                launches(want)
                ctx.synchronize()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(z, want), rep
        launches(z)                                                           # plain launches again after the graph
        ctx.synchronize()
        assert torch.equal(z, want)


def test_point_cloud_text_formatted_on_the_device(api, oracle, synth, torch_cuda, tmp_path):
    """slx_format_points_text / slx_get_point_cloud_text (csrc/slx_text.hip): the text of CCalculation::Result (R/CCalculation.cpp:351-353,
    `file << x << ' ' << y << ' ' << z << endl`) formatted on the device must be the bytes of the host writer (slx_write_point_cloud_text)
    and of "%g": magnitudes across the formatter's whole range, both notations and their borders (X = -5, -4, 5, 6), ties of the sixth
    digit on exactly representable values (round half to even), carries into a new digit (999999.5 -> 1e+06, 99999.95 -> 100000),
    zeros of both signs, lengths that put the workgroups' boundaries at every alignment, a last workgroup with one point; a value
    outside the range (tiny, huge, NaN, infinity) makes the call fail with SLX_ERR_UNAVAILABLE so that the caller formats on the host.
    Then the cloud of a decoded frame and of a tracked frame at 1920 x 1200 against the host writer's file."""
    torch = torch_cuda
    rng = np.random.default_rng(17)
    spec = synth.make_spec("C4")
    H, W = spec["height"], spec["width"]

    def fmt(a):
        return ("".join("%g %g %g\n" % tuple(p) for p in a)).encode()

    with api.Context(spec, aux=("U",)) as ctx:
        cases = []
        a = (rng.random((70001, 3)) - 0.3) * 1500.0
        a[::5] = rng.standard_normal((a[::5].shape[0], 3)) * 10.0 ** rng.integers(-4, 14, size=(a[::5].shape[0], 1))
        a[np.abs(a) < 1e-5] = 0.0
        a[:8] = [[0.0, -0.0, 1.0], [999999.5, 999999.4, 0.0001], [0.00009999995, 123456.5, 1234565.0], [100000.5, 99999.95, 123.4565],
                 [0.5, 2.5, 1.5], [1e-5, 9.99999e-5, 0.000123456], [999999.0, 1e6, 9.999995e14], [-1e-5, -123456.5, -0.0001234565]]
        cases.append(a)
        # ties: integers and halves are exact in binary: x.5 at the sixth digit must round to even
        t = np.array([[100000.5 + 2 * k, 100001.5 + 2 * k, 12345.65 + k] for k in range(300)])
        cases.append(np.concatenate([t, t / 1024.0, t * 4096.0, -t]))
        cases.append(np.array([[1.0, 2.0, 3.0]]))                                # one point
        cases.append((rng.random((1025, 3)) - 0.5) * 10.0 ** rng.integers(-5, 15, size=(1025, 1)))   # one workgroup and one point
        cases[-1][np.abs(cases[-1]) < 1e-5] = 0.0
        for k, a in enumerate(cases):
            a = np.ascontiguousarray(a, dtype=np.float64)
            dev = torch.from_numpy(a).cuda()
            torch.cuda.synchronize()
            got = ctx.format_points_text(dev)
            want = fmt(a)
            assert len(got) == len(want) and got == want, (k, [(i, g, w) for i, (g, w) in enumerate(zip(got.split(b"\n"), want.split(b"\n"))) if g != w][:3])
            path = str(tmp_path / "host.txt")
            api.write_point_cloud_text(path, a)
            assert open(path, "rb").read() == got, k
        assert ctx.format_points_text(torch.empty((0, 3), dtype=torch.float64, device="cuda")) == b""
        # a cloud of more than 4 Mi points (4 096 workgroups): the characters' launch then takes the bytes in front of every run of 1 024
        # workgroups from a small launch in between instead of adding up all lengths before it (linear, not quadratic, in the points)
        big = (rng.random((4_400_000, 3)) - 0.4) * 10.0 ** rng.integers(-3, 7, size=(4_400_000, 1))
        big[np.abs(big) < 1e-5] = 0.0
        dev = torch.from_numpy(big).cuda()
        torch.cuda.synchronize()
        got = ctx.format_points_text(dev)
        path = str(tmp_path / "big.txt")
        api.write_point_cloud_text(path, big)
        assert open(path, "rb").read() == got
        del dev, big, got
        for bad in (1e-7, -3e-6, 1e15, -2.5e200, np.nan, np.inf, -np.inf, 5e-324):
            a = np.ones((2050, 3))
            a[2049, 1] = bad
            dev = torch.from_numpy(a).cuda()
            torch.cuda.synchronize()
            with pytest.raises(api.SlxError) as e:
                ctx.format_points_text(dev)
            assert e.value.code == api.ERR_UNAVAILABLE, bad
            assert ctx.format_points_text(dev[:2049]) == fmt(a[:2049])          # the next call is not affected
        # a decoded frame's cloud, then a tracked frame's
        ph, gr, _ = synth.render(spec, "sphere", noise_sigma=2.0)
        imgs = dyna_images(H, W, 2, seed=4)
        ctx.set_frames(ph, gr)
        ctx.decode()
        for step in range(2):
            if step == 1:
                ctx.track_begin(imgs[0])
                ctx.track_next(imgs[1])
            text, n = ctx.get_point_cloud_text()
            cloud = ctx.get_point_cloud()
            assert n == len(cloud) and 0 < n < H * W
            path = str(tmp_path / "cloud.txt")
            api.write_point_cloud_text(path, cloud)
            assert open(path, "rb").read() == text, step
            assert text[:200] == fmt(cloud[:40])[:200]
            # the pipeline (cloud -> lengths -> piece offsets, one wait, then the characters piece by piece beside their copies) in
            # every number of pieces, against the plain sequence (1 piece); also with the count + write launches in front
            for pieces, passes in ((1, 0), (2, 0), (3, 0), (16, 0), (8, 2), (0, 0)):
                ctx.set_tuning(text_pieces=pieces, cloud_passes=passes)
                t2, n2 = ctx.get_point_cloud_text()
                assert n2 == n and t2 == text, (step, pieces, passes)
            ctx.set_tuning(text_pieces=0, cloud_passes=0)


def test_point_cloud_text_in_the_dialect_of_the_reference_as_built(api, oracle, synth, torch_cuda, tmp_path):
    """`ostream << double` means different bytes on the two runtimes: libstdc++ / glibc print "5e-05" and '\\n', the reference AS BUILT
    (MSVC 2013 runtime, text-mode stream; R/CCalculation.cpp:323-357) prints "5e-005" and CR LF.  SLX_TEXT_MSVC2013 on the device
    (slx_set_text_dialect) and on the host (slx_write_point_cloud_text_ex) must give the libstdc++ text with every exponent padded to
    three digits and CR LF line ends -- on values in exponent notation of both signs, at the notation borders, and on a decoded frame's
    cloud, whose x next to cx is in exponent notation.  (Unpinned like the rest: the reference ships no output file.)"""
    import re
    torch = torch_cuda
    rng = np.random.default_rng(23)

    def msvc(a):
        text = "".join("%g %g %g\r\n" % tuple(p) for p in a)
        return re.sub(r"e([+-])(\d\d)(?!\d)", r"e\g<1>0\2", text).encode()

    spec = synth.make_spec("C4")
    with api.Context(spec) as ctx:
        a = (rng.random((5000, 3)) - 0.5) * 10.0 ** rng.integers(-5, 15, size=(5000, 1))
        a[np.abs(a) < 1e-5] = 0.0
        a[:6] = [[5e-5, -5e-5, 1.5e-5], [1e6, 1.25e14, -9.99999e-5], [999999.5, 999999.4, 0.0001], [0.00009999995, 1e-5, 123456.5], [0.0, -0.0, 1.0], [9.999995e14, -1e6, 2.5]]
        dev = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        torch.cuda.synchronize()
        plain = ctx.format_points_text(dev)
        assert plain == ("".join("%g %g %g\n" % tuple(p) for p in a)).encode()
        ctx.set_text_dialect(api.TEXT_MSVC2013)
        got = ctx.format_points_text(dev)
        assert got == msvc(a), [(i, g, w) for i, (g, w) in enumerate(zip(got.split(b"\r\n"), msvc(a).split(b"\r\n"))) if g != w][:3]
        assert got.startswith(b"5e-005 -5e-005 1.5e-005\r\n1e+006 1.25e+014 -9.99999e-005\r\n1e+006 999999 0.0001\r\n0.0001 1e-005 123456\r\n0 -0 1\r\n")
        path = str(tmp_path / "msvc.txt")
        api.write_point_cloud_text(path, a, dialect=api.TEXT_MSVC2013)
        assert open(path, "rb").read() == got
        with pytest.raises(api.SlxError):
            ctx.set_text_dialect(5)
    # a decoded frame whose principal point lies 2e-4 px from a pixel column: that column's x = z (u - cx) / fu is ~4e-5, exponent notation
    spec2 = dict(spec)
    cal = dict(spec["calib"])
    cam = list(cal["cam"])
    cam[2] = 960.0002
    cal["cam"] = cam
    spec2["calib"] = cal
    ph, gr, _ = synth.render(spec2, "sphere", noise_sigma=2.0)
    with api.Context(spec2) as ctx:
        ctx.set_text_dialect(api.TEXT_MSVC2013)
        ctx.set_frames(ph, gr)
        ctx.decode()
        text, n = ctx.get_point_cloud_text()
        cloud = ctx.get_point_cloud()
        assert n == len(cloud) and text == msvc(cloud) and text.count(b"e-005") > 500
        ctx.set_text_dialect(api.TEXT_LIBSTDCXX)
        text2, _ = ctx.get_point_cloud_text()
        assert text2 == text.replace(b"\r\n", b"\n").replace(b"e-005", b"e-05")


def test_tracked_frame_cloud_full_size(api, oracle, synth, torch_cuda):
    """The reference writes a cloud after every dynamic frame (main loop: CalculateOther, Result): 1920 x 1200, three tracked frames in a
    row, the cloud of each into host memory (count first), as the context's pinned view, into a device buffer for every pixel and into
    one exactly as large as the cloud -- by the fused launch and by count + write, against the oracle."""
    import ctypes
    torch = torch_cuda
    spec = synth.make_spec("C4")
    H, W = spec["height"], spec["width"]
    ph, gr, _ = synth.render(spec, "sphere", noise_sigma=2.0)
    imgs = dyna_images(H, W, 4, seed=11)
    dev = torch.full((H * W, 3), -7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with api.Context(spec, aux=("U",)) as ctx:
        ctx.set_frames(ph, gr)
        ctx.decode()
        ctx.track_begin(imgs[0])
        for f in (1, 2, 3):
            ctx.track_next(imgs[f])
            ref = oracle.point_cloud(spec, ctx.get_depth())
            assert 0 < len(ref) < H * W
            for passes in (0, 1, 2):
                ctx.set_tuning(cloud_passes=passes)
                got = ctx.get_point_cloud()                      # count only, then into host memory
                assert got.shape == ref.shape and np.array_equal(got, ref, equal_nan=True), (f, passes)
            ctx.set_tuning(cloud_passes=0)
            assert np.array_equal(ctx.get_point_cloud_view(), ref, equal_nan=True), f
            n = ctypes.c_size_t(0)
            assert api.lib().slx_get_point_cloud(ctx._h, dev.data_ptr(), H * W, ctypes.byref(n), api.MEM_DEVICE) == 0
            assert n.value == len(ref) and np.array_equal(dev[:n.value].cpu().numpy(), ref, equal_nan=True), f
            tight = torch.full((len(ref) + 1, 3), -7.0, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()                             # (the fill runs on torch's stream, the library on its own)
            assert api.lib().slx_get_point_cloud(ctx._h, tight.data_ptr(), len(ref), ctypes.byref(n), api.MEM_DEVICE) == 0
            assert n.value == len(ref) and np.array_equal(tight[:-1].cpu().numpy(), ref, equal_nan=True) and bool((tight[-1] == -7.0).all()), f


def test_dynamic_frames_device_images_full_size(api, oracle, synth, torch_cuda):
    torch = torch_cuda
    spec = synth.make_spec("REF")                             # 1280 x 1024, the reference's compiled-in camera
    H, W = spec["height"], spec["width"]
    ph, gr, _ = synth.render(spec, "sphere", noise_sigma=2.0)
    imgs = dyna_images(H, W, 3, seed=3)
    dev = [torch.from_numpy(np.pad(i, ((0, 0), (0, 64)))).cuda()[:, :W] for i in imgs]     # padded rows, borrowed
    torch.cuda.synchronize()
    ref0 = oracle.pipeline(spec, ph, gr, want=("z", "U"), threads=8)
    with api.Context(spec, aux=("U",)) as ctx:
        ctx.set_frames(ph, gr)
        ctx.decode()
        ctx.track_begin(dev[0])
        sw0, sb0 = oracle.strip_regression(imgs[0])
        U, z_prev = ref0["U"], ref0["z"]
        for f in (1, 2):
            ctx.track_next(dev[f])
            sw1, sb1 = oracle.strip_regression(imgs[f])
            U = U + oracle.delta_p(sw0, sb0, sw1, sb1).astype(np.float64)
            tri = oracle.triangulate(spec, U)
            assert np.array_equal(ctx.get_output("U"), U) and np.array_equal(ctx.get_depth(), tri["z"], equal_nan=True)
            assert np.array_equal(ctx.get_output("deltaZ"), tri["z"] - z_prev, equal_nan=True)
            sw0, sb0, z_prev = sw1, sb1, tri["z"]


def test_dynamic_frames_fed_through_the_pinned_buffer(api, oracle, synth):
    """slx_track_image_buffer: images written straight into the staging slot (no library copy), mixed with ordinary host
    images and a strided one, give the oracle's frames; the two slots alternate."""
    h, w = 64, 200
    spec = small_spec(synth, "C1x4", w, h)
    ph, gr, _ = synth.render(spec, "tilted", noise_sigma=2.0)
    ref0 = oracle.pipeline(spec, ph, gr, want=("z", "U"))
    imgs = dyna_images(h, w, 8, seed=77)
    with api.Context(spec, aux=("U",)) as ctx:
        ctx.set_frames(ph, gr)
        ctx.decode()
        buf = ctx.track_image_buffer()
        assert buf.shape == (h, w) and buf.dtype == np.uint8
        buf[:] = imgs[0]
        ctx.track_begin(buf)
        seen = {buf.ctypes.data}
        sw0, sb0 = oracle.strip_regression(imgs[0])
        U = ref0["U"]
        for f in range(1, 8):
            if f % 3 == 0:
                ctx.track_next(imgs[f])                              # the library copies
            elif f == 4:
                wide = np.zeros((h, w + 24), dtype=np.uint8)
                wide[:, :w] = imgs[f]
                ctx.track_next(wide[:, :w])                          # strided host image
            else:
                buf = ctx.track_image_buffer()
                again = ctx.track_image_buffer()                     # asking twice hands out the same slot
                assert again.ctypes.data == buf.ctypes.data
                seen.add(buf.ctypes.data)
                buf[:] = imgs[f]
                ctx.track_next(buf)
            sw1, sb1 = oracle.strip_regression(imgs[f])
            U = U + oracle.delta_p(sw0, sb0, sw1, sb1).astype(np.float64)
            assert np.array_equal(ctx.get_output("U"), U), f
            assert np.array_equal(ctx.get_depth(), oracle.triangulate(spec, U)["z"], equal_nan=True), f
            sw0, sb0 = sw1, sb1
        assert len(seen) == 2
    with api.Context(spec) as ctx:                                   # no U plane: no tracker, no buffer
        with pytest.raises(api.SlxError) as e:
            ctx.track_image_buffer()
        assert e.value.code == api.ERR_UNAVAILABLE


@pytest.mark.parametrize("shape,window", [((64, 200), 21), ((130, 256), 21), ((48, 332), 9)])
def test_dynamic_frames_in_batches(api, oracle, synth, torch_cuda, shape, window):
    """slx_track_next_batch / slx_track_stage_frames: k camera images per transfer.  The oracle's loop frame by frame
    (R/CCalculation.cpp:789-892, 595-663, 666-785); every frame's deltaZ through deltaz_all (device and host), the context's
    outputs after a batch are the last frame's; batches of different sizes (the slabs grow), images from the pinned slab, strided
    host images, device images, and the stage-then-step form CCalculation::CalculateOther uses (a point cloud per frame)."""
    torch = torch_cuda
    h, w = shape
    spec = small_spec(synth, "C1x4", w, h)
    ph, gr, _ = synth.render(spec, "tilted", noise_sigma=2.0)
    ref0 = oracle.pipeline(spec, ph, gr, want=("z", "U"))
    n = 14
    imgs = dyna_images(h, w, n, seed=5 * h + w)
    # the oracle's frames
    want = []
    sw0, sb0 = oracle.strip_regression(imgs[0], window)
    U, z_prev = ref0["U"], ref0["z"]
    for f in range(1, n):
        sw1, sb1 = oracle.strip_regression(imgs[f], window)
        dP = oracle.delta_p(sw0, sb0, sw1, sb1)
        U = U + dP.astype(np.float64)
        tri = oracle.triangulate(spec, U)
        want.append({"stripW": sw1, "stripB": sb1, "deltaP": dP, "U": U, "z": tri["z"], "x": tri["x"], "y": tri["y"], "deltaZ": tri["z"] - z_prev})
        sw0, sb0, z_prev = sw1, sb1, tri["z"]

    def check_last(ctx, f):
        for name, ref in want[f - 1].items():
            assert np.array_equal(ctx.get_output(name), ref, equal_nan=True), (f, name)

    with api.Context(spec, aux=("U", "x", "y")) as ctx:
        ctx.set_frames(ph, gr)
        ctx.decode()
        with pytest.raises(api.SlxError) as e:
            ctx.track_next_batch(np.stack(imgs[1:3]))                # before slx_track_begin
        assert e.value.code == api.ERR_NOT_CONFIGURED
        ctx.track_begin(imgs[0], window=window)
        # frames 1-3: one host batch, every deltaZ into a device array
        dz = torch.full((3, h, w), -7.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ctx.track_next_batch(np.stack(imgs[1:4]), dz)
        ctx.synchronize()
        check_last(ctx, 3)
        for k in range(3):
            assert np.array_equal(dz[k].cpu().numpy(), want[k]["deltaZ"], equal_nan=True), k
        # frames 4-8: a larger batch (the slabs grow) written straight into the pinned slab, deltaZ into host memory
        slab = ctx.track_frames_buffer(5)
        assert slab.shape == (5, h, w)
        slab[:] = np.stack(imgs[4:9])
        dzh = np.full((5, h, w), -7.0)
        ctx.track_next_batch(slab, dzh)
        check_last(ctx, 8)
        for k in range(5):
            assert np.array_equal(dzh[k], want[3 + k]["deltaZ"], equal_nan=True), k
        # frame 9: a batch of one strided host image, no deltaZ collection
        wide = np.zeros((1, h, w + 40), dtype=np.uint8)
        wide[0, :, :w] = imgs[9]
        ctx.track_next_batch(wide[:, :, :w])
        check_last(ctx, 9)
        # frames 10-11: device images, borrowed
        dev = torch.from_numpy(np.stack(imgs[10:12])).cuda()
        torch.cuda.synchronize()
        ctx.track_next_batch(dev)
        ctx.synchronize()
        check_last(ctx, 11)
        # frames 12-13: one transfer, then frame by frame from the device slab, the cloud of each frame in between
        base = ctx.track_stage_frames(np.stack(imgs[12:14]))
        for k, f in enumerate((12, 13)):
            ctx.track_next_device(base + k * h * w)
            check_last(ctx, f)
            assert np.array_equal(ctx.get_point_cloud(), oracle.point_cloud(spec, want[f - 1]["z"]))
        with pytest.raises(api.SlxError) as e:
            ctx.track_next_batch(np.zeros((0, h, w), dtype=np.uint8))
        assert e.value.code == api.ERR_INVALID_ARG
        # a pointer into the slab handed out for 5 frames, passed back for 9: the slabs would be freed and regrown under it -- refused,
        # nothing staged, and the tracker carries on from where it was
        small = ctx.track_frames_buffer(5)
        nine = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint8 * (5 * h * w)).from_address(small.ctypes.data)).reshape(5, h, w)
        with pytest.raises(api.SlxError) as e:
            dev_out = np.ctypeslib.ctypes.c_void_p()
            ctx._check(api.lib().slx_track_stage_frames(ctx._h, nine.ctypes.data, w, h * w, 9, np.ctypeslib.ctypes.byref(dev_out)))
        assert e.value.code == api.ERR_INVALID_ARG and "handed out for 5 frames" in str(e.value)
        check_last(ctx, 13)


@pytest.mark.parametrize("shape", [(70, 200), (64, 64), (129, 65), (5, 700)])
def test_point_cloud_of_a_batch_plane(api, oracle, synth, torch_cuda, shape):
    """slx_point_cloud_of_depth: the cloud (CCalculation::Result's data) of every frame-set of a batch decode, straight from
    the batch's depth planes, into host memory and into a device buffer; a context that never ran a single-set decode;
    tiles ragged in both directions; a plane with no point inside the FOV; refused inputs."""
    torch = torch_cuda
    h, w = shape
    spec = small_spec(synth, "C4", w, h)
    sets = [synth.render(spec, kind, seed=5 + i, noise_sigma=2.0)[:2] for i, kind in enumerate(("tilted", "sphere", "tilted"))]
    zs = oracle.pipeline(spec, *sets[0], want=("z",))["z"]
    spec["fov_min"], spec["fov_max"] = float(np.percentile(zs[zs > 0], 20)), float(np.percentile(zs[zs > 0], 85))   # a partial cloud
    want = [oracle.pipeline(spec, ph, gr, want=("z",))["z"] for ph, gr in sets]
    phase = torch.from_numpy(np.stack([ph for ph, _ in sets])).cuda()
    z = torch.full((4, h, w), -1.0, dtype=torch.float64, device="cuda")      # plane 3: never decoded, every depth outside the FOV
    z[3] = spec["fov_max"] + 1.0
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with api.Context(spec) as ctx:
        ctx.decode_batch(3, phase, None, z[:3], stream=side.cuda_stream)     # the cloud orders itself after this launch
        dev = torch.full((h * w, 3), np.nan, dtype=torch.float64, device="cuda")
        for i in range(3):
            ref = oracle.point_cloud(spec, want[i])
            got = ctx.point_cloud_of_depth(z[i])
            assert got.shape == ref.shape and np.array_equal(got, ref), i
            n = ctx.point_cloud_of_depth(z[i], out=dev)
            assert n == len(ref) and np.array_equal(dev[:n].cpu().numpy(), ref), i
        assert 0 < len(oracle.point_cloud(spec, want[0])) < h * w
        assert ctx.point_cloud_of_depth(z[3]).shape == (0, 3)
        small = torch.empty((1, 3), dtype=torch.float64, device="cuda")
        if len(oracle.point_cloud(spec, want[0])) > 1:
            with pytest.raises(api.SlxError) as e:
                ctx.point_cloud_of_depth(z[0], out=small)                   # too small: refused, the count is still reported
            assert e.value.code == api.ERR_INVALID_ARG
        n = C_size(0)
        assert api.lib().slx_point_cloud_of_depth(ctx._h, None, None, 0, n, api.MEM_DEVICE) == api.ERR_INVALID_ARG
        assert api.lib().slx_point_cloud_of_depth(ctx._h, z.data_ptr() + 4, None, 0, n, api.MEM_DEVICE) == api.ERR_INVALID_ARG
    pspec = {"width": w, "height": h, "mode": synth.MODE_PHASE_ONLY, "n_freq": 1, "n_steps": 4, "periods": [40]}   # no depth, no cloud
    with api.Context(pspec) as ctx:
        n = C_size(0)
        assert api.lib().slx_point_cloud_of_depth(ctx._h, z.data_ptr(), None, 0, n, api.MEM_DEVICE) == api.ERR_UNAVAILABLE


def C_size(v):
    import ctypes
    return ctypes.byref(ctypes.c_size_t(v))


# ------------------------------------------------------------------ sharding on the device
def test_row_tiles_and_frameset_shards_on_gpu(api, oracle, synth, shard):
    spec = small_spec(synth, "C3", 128, 50)
    ph, gr = synth.random_planes(spec, seed=21)
    full = oracle.pipeline(spec, ph, gr, want=("z", "y"))
    for world in (2, 8):
        parts = []
        for rank in range(world):
            tile, lo, hi = shard.row_tile_spec(spec, world, rank)
            parts.append(api.decode_frameset(tile, ph[:, lo:hi], gr[:, lo:hi], want=("z", "y")))
        for w in ("z", "y"):
            assert np.array_equal(np.concatenate([p[w] for p in parts]), full[w], equal_nan=True), (world, w)


def test_rccl_backend_single_rank_collectives(tmp_path):
    """bench.py's multi-rank calls (init with device_id, barrier, all_reduce MAX, gather into views of one buffer,
    all_gather_into_tensor) on the RCCL backend.  One rank: the box has one GPU; the 2-rank logic runs on gloo in
    tests/test_shard_gloo.py.  In a child process so that the process group does not leak into this one."""
    import subprocess
    import sys
    code = r'''
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.tolist() == [1.5, 2.5]
local = torch.arange(2 * 3 * 4, dtype=torch.float64, device=dev).reshape(2, 3, 4)
out = torch.empty((2, 3, 4), dtype=torch.float64, device=dev)
dist.gather(local, list(out.split(2, dim=0)), dst=0)
assert torch.equal(out, local)
out2 = torch.empty_like(out); dist.all_gather_into_tensor(out2, local)
assert torch.equal(out2, local)
dist.destroy_process_group()
print("rccl ok")
'''
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout + r.stderr


# ------------------------------------------------------------------ full batch size, size-independent properties
def test_full_size_batch_properties(api, oracle, synth, torch_cuda):
    """BASELINE configuration 4's per-GPU share: 32 frame-sets of 1920x1200, 3 x 4-step, in one launch.
    Property: decoding is per-frame-set and per-pixel independent, so (a) a batch of copies of
    one frame-set gives identical maps, each equal to the oracle's; (b) permuting the sets of a
    batch permutes the outputs; (c) a second launch reproduces the first bit for bit."""
    torch = torch_cuda
    spec = synth.make_spec("C4")
    H, W = spec["height"], spec["width"]
    ph0, _, _ = synth.render(spec, "sphere", seed=77, noise_sigma=2.0)
    ph1, _ = synth.random_planes(spec, seed=78)
    ref0 = oracle.pipeline(spec, ph0, None, want=("z",), threads=8)["z"]
    ref1 = oracle.pipeline(spec, ph1, None, want=("z",), threads=8)["z"]
    n_sets = 32
    batch = torch.empty((n_sets, 12, H, W), dtype=torch.uint8, device="cuda")
    d0, d1 = torch.from_numpy(ph0).cuda(), torch.from_numpy(ph1).cuda()
    odd = [s for s in range(n_sets) if s % 5 == 2]
    for s in range(n_sets):
        batch[s] = d1 if s in odd else d0
    z = torch.empty((n_sets, H, W), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()                  # the context's stream does not order itself against torch's stream
    with api.Context(spec) as ctx:
        ctx.decode_batch(n_sets, batch, None, z)
        ctx.synchronize()
        assert ctx.last_kernel() == "slx_stream_kernel<3>: resident waves, 2-row items from queues", ctx.last_kernel()   # the headline launch (bench.py)
        first = z.clone()
        r0, r1 = torch.from_numpy(ref0).cuda(), torch.from_numpy(ref1).cuda()
        for s in range(n_sets):
            want = r1 if s in odd else r0
            assert torch.equal(torch.nan_to_num(z[s], nan=-7.0), torch.nan_to_num(want, nan=-7.0)), s
        z.fill_(-1.0)
        torch.cuda.synchronize()              # or the fill could land after the decode
        ctx.decode_batch(n_sets, batch, None, z)
        ctx.synchronize()
        assert torch.equal(torch.nan_to_num(z, nan=-7.0), torch.nan_to_num(first, nan=-7.0))


def test_c4_batch_on_the_strip_kernel_16_row_items_woven_8(api, oracle, synth, torch_cuda):
    """The launch plan that serves the reference's real batch output -- x, y beside z for every frame (R/CCalculation.cpp:756-771,
    :666-785) -- at BASELINE configuration 4's per-GPU size: 32 frame-sets of 1920 x 1200, 3 x 4-step, on slx_strip_kernel with
    16-row items woven 8 rows to a row group (the planner's own choice for >= 26 such frame-sets once the stream kernel is out:
    it is depth-only).  1200 rows are not a multiple of the 128-row groups: the tail tier covers 192 rows for the last 176, so the
    last row group of every frame-set walks 16 rows past the tile (DMA clamped to the last row, stores dropped by the descriptor's
    range check) -- the geometry round 4's tools/weave_check.py reported a mismatch on before its fills were ordered against the
    decode stream.  Four legs: {automatic tuning, weave=8 + strip_rows=16 forced} x {depth only, with x, y, U, k
    through slx_decode_batch_ex}; 4 distinct frame-sets (a rendered scene and three of unstructured bytes) spread over the 32; EVERY
    frame-set of every leg against the oracle, bit for bit (contract: 1e-4 mm RMS, fringe orders exact); slx_last_kernel must
    name the instantiation and the item geometry.  Outputs are pre-filled with -7: a row the launch skipped would show."""
    torch = torch_cuda
    spec = synth.make_spec("C4")
    H, W, n_sets = spec["height"], spec["width"], 32
    assert (W, H) == (1920, 1200)
    scenes = [synth.render(spec, "sphere", seed=501, noise_sigma=2.0)[0]] + [synth.random_planes(spec, seed=502 + s)[0] for s in range(3)]
    names = ("z", "x", "y", "U", "k")
    refs = [oracle.pipeline(spec, ph, None, want=names, threads=8) for ph in scenes]
    dev_ref = [{w: torch.from_numpy(np.ascontiguousarray(r[w])).cuda() for w in names} for r in refs]
    which = [(5 * s + s // 7) % 4 for s in range(n_sets)]            # every scene several times, no period that divides a tier
    assert set(which) == {0, 1, 2, 3} and which[0] != which[-1]
    dev_scenes = [torch.from_numpy(ph).cuda() for ph in scenes]
    batch = torch.empty((n_sets, 12, H, W), dtype=torch.uint8, device="cuda")
    for s in range(n_sets):
        batch[s] = dev_scenes[which[s]]
    del dev_scenes

    def same(a, b):
        if a.dtype == torch.float64:
            return torch.equal(a.view(torch.int64), b.view(torch.int64)) or torch.equal(torch.nan_to_num(a, nan=-7.5), torch.nan_to_num(b, nan=-7.5))
        return torch.equal(a, b)

    with api.Context(spec) as ctx:
        for tune in (dict(stream=1), dict(stream=1, weave=8, strip_rows=16)):
            ctx.set_tuning(weave=0, strip_rows=0, **{k: v for k, v in tune.items() if k == "stream"})
            ctx.set_tuning(**tune)
            for aux in (False, True):
                outs = {"z": torch.full((n_sets, H, W), -7.0, dtype=torch.float64, device="cuda")}
                if aux:
                    for w in ("x", "y", "U"):
                        outs[w] = torch.full((n_sets, H, W), -7.0, dtype=torch.float64, device="cuda")
                    outs["k"] = torch.full((n_sets, 2, H, W), -7, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()          # the fills run on torch's stream, the decode on the context's own: order them
                if aux:
                    ctx.decode_batch_ex(n_sets, batch, None, **outs)
                else:
                    ctx.decode_batch(n_sets, batch, None, outs["z"])
                ctx.synchronize()
                meant = "slx_strip_kernel<3, 3, 0, 4, %s>: 16-row items, 8 rows per row group" % ("true" if aux else "false")
                assert ctx.last_kernel() == meant, (ctx.last_kernel(), tune, aux)
                for s in range(n_sets):
                    for w in outs:
                        assert same(outs[w][s], dev_ref[which[s]][w]), "frame-set %d (scene %d): %s differs from the oracle (%s, aux=%s)" % (s, which[s], w, tune, aux)
                del outs


# ------------------------------------------------------------------ frame ingest pipeline (host frames, pinned slots)
@pytest.mark.parametrize("name,shape", [("C3", (37, 130)), ("C2", (50, 64)), ("C1", (20, 33)), ("C5", (24, 96))])
def test_ingest_pipe_matches_oracle_in_submit_order(api, oracle, synth, name, shape):
    """slx_pipe_*: 3 slots x 2 frame-sets, 9 frame-sets pushed with the slots kept full; every result must be the
    oracle's map of the frame-set submitted in that position (results come back oldest first), bit for bit."""
    h, w = shape
    spec = small_spec(synth, name, w, h)
    primary = "pix" if spec["mode"] == 0 else "z"
    sets = [synth.random_planes(spec, seed=100 + i) for i in range(9)]
    want = []
    for ph, gr in sets:
        o = oracle.pipeline(spec, ph, gr, want=(primary,))[primary]
        want.append(o[0] if primary == "pix" else o)
    with api.Context(spec) as ctx:
        pipe = api.Pipe(ctx, slots=3, sets_per_slot=2)
        assert pipe.pitch == (w + 3) // 4 * 4 and pipe.n_planes == sum(x.shape[0] for x in sets[0] if x is not None)
        groups = [sets[0:2], sets[2:4], sets[4:6], sets[6:8], sets[8:9]]     # the last slot is only half full
        got, in_flight = [], 0
        for g in groups:
            if in_flight == 3:                                                  # all slots busy: take the oldest result
                got.extend(np.array(m) for m in pipe.collect())
                in_flight -= 1
            buf = pipe.acquire()
            buf[...] = 0xA5                                                     # stale bytes in the padding / unused sets
            for i, (ph, gr) in enumerate(g):
                planes = np.concatenate([x for x in (ph, gr) if x is not None])
                buf[i, :, :, :w] = planes
            pipe.submit(len(g))
            in_flight += 1
        while in_flight:
            got.extend(np.array(m) for m in pipe.collect())
            in_flight -= 1
        with pytest.raises(api.SlxError):
            pipe.collect()                                                      # nothing left
        pipe.close()
    assert len(got) == 9
    for i in range(9):
        assert np.array_equal(got[i], want[i], equal_nan=True), i


def test_ingest_pipe_slot_rules(api, synth):
    spec = small_spec(synth, "C2", 32, 8)
    with api.Context(spec) as ctx:
        with pytest.raises(api.SlxError):
            api.Pipe(ctx, slots=1)
        pipe = api.Pipe(ctx, slots=2, sets_per_slot=1, host_result=False)
        pipe.acquire()
        with pytest.raises(api.SlxError):
            pipe.acquire()                                                      # one acquisition at a time
        with pytest.raises(api.SlxError):
            pipe.submit(2)                                                      # more sets than the slot holds
        pipe.submit(1)
        pipe.acquire()[...] = 0
        pipe.submit(1)
        with pytest.raises(api.SlxError):
            pipe.acquire()                                                      # both slots in flight
        dev, n = pipe.collect()
        assert dev and n == 1
        pipe.acquire()                                                          # the collected slot is reusable
        pipe.submit(1)
        pipe.collect()
        pipe.collect()
        pipe.close()


# ------------------------------------------------------------------ the C++ mirror of the reference classes
def test_cpp_host_loop(tmp_path, oracle, synth, golden_dir):
    """tests/cpp/dynaframe_host_loop.cpp drives slx::CDecodeGray / CDecodePhase / CCalculation the way
    CCalculation::FillFirstProjectorU + CalculateFirst do (R/CCalculation.cpp:171-206, :525-592)."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "dynaframe_host_loop")
    from conftest import _ensure_built
    _ensure_built()
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    W, H, PW = 320, 200, 1280
    spec = small_spec(synth, "C1x4", W, H)
    ph, gr, _ = synth.render(spec, "sphere", seed=3, noise_sigma=2.0)
    rows = json.load(open(os.path.join(golden_dir, "vGrayCode_rows.json")))["rows"]
    code_dir = str(tmp_path) + "/"
    with open(code_dir + "vGrayCode.txt", "w") as f:          # the reference's file format: "bin gray" per line
        f.write("\n".join("%d %d" % (b, g) for b, g in rows) + "\n")
    np.concatenate([gr.reshape(-1), ph.reshape(-1)]).tofile(code_dir + "in.bin")
    subprocess.check_call([exe, code_dir + "in.bin", code_dir + "out.bin", str(W), str(H), str(PW), code_dir, "vGrayCode.txt"])
    out = np.fromfile(code_dir + "out.bin", dtype=np.float64).reshape(6, H, W)
    ref = oracle.pipeline(spec, ph, gr, want=("gray", "pix", "z", "x", "y", "U"))
    # CCalculation::Result: the text point cloud, default ostream formatting == "%g"
    pts = oracle.point_cloud(spec, ref["z"])
    want_txt = "".join("%g %g %g\n" % tuple(p) for p in pts)
    assert open(code_dir + "out.bin.txt").read() == want_txt
    for i, w in enumerate(("gray", "pix", "z", "x", "y", "U")):
        r = ref[w][0] if w == "pix" else ref[w]
        assert np.array_equal(out[i], r, equal_nan=True), w


def test_cpp_data_directory(tmp_path, oracle, synth, golden_dir):
    """The reference's main() over a DynaFrame data directory: parameters.yml, iFrame/v{Gray,Phase}Cam<i>.bmp,
    Patterns/vGrayCode.txt in, PointCloud/iFrame.txt out (tests/cpp/dynaframe_data_dir.cpp)."""
    import subprocess
    from conftest import ROOT
    from dynaframe_files import write_bmp, write_calibration_yaml
    exe = os.path.join(ROOT, "tests", "cpp", "dynaframe_data_dir")
    from conftest import _ensure_built
    _ensure_built()
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    W, H, PW = 256, 130, 1280
    spec = small_spec(synth, "C1x4", W, H)
    ph, gr, _ = synth.render(spec, "sphere", seed=5, noise_sigma=2.0)
    d = str(tmp_path)
    for sub in ("g/iFrame", "Patterns", "PointCloud"):
        os.makedirs(os.path.join(d, sub))
    cal = spec["calib"]
    write_calibration_yaml(os.path.join(d, "parameters.yml"), cal["cam"], cal["pro"], cal["rot"], cal["trans"])
    rows = json.load(open(os.path.join(golden_dir, "vGrayCode_rows.json")))["rows"]
    with open(os.path.join(d, "Patterns", "vGrayCode.txt"), "w") as f:
        f.write("\n".join("%d %d" % (b, g) for b, g in rows) + "\n")
    for i in range(12):
        write_bmp(os.path.join(d, "g/iFrame/vGrayCam%d.bmp" % i), gr[i], bits=8 if i % 2 else 24, top_down=bool(i % 3 == 0))
    for i in range(4):
        write_bmp(os.path.join(d, "g/iFrame/vPhaseCam%d.bmp" % i), ph[i], bits=8)
    os.makedirs(os.path.join(d, "g/cFrame"))
    dyn = dyna_images(H, W, 4, seed=8)
    for i, img in enumerate(dyn):
        write_bmp(os.path.join(d, "g/cFrame/dynaCam%d.bmp" % i), img, bits=8)
    out = subprocess.check_output([exe, d, "g", str(PW), str(spec["fov_min"]), str(spec["fov_max"])]).decode()
    assert out.startswith("ok %d x %d" % (W, H)) and "dynamic frames 3" in out
    ref = oracle.pipeline(spec, ph, gr, want=("z",))
    z = np.fromfile(os.path.join(d, "z.bin"), dtype=np.float64).reshape(H, W)
    assert np.array_equal(z, ref["z"], equal_nan=True)
    pts = oracle.point_cloud(spec, ref["z"])
    assert open(os.path.join(d, "PointCloud", "iFrame.txt")).read() == "".join("%g %g %g\n" % tuple(p) for p in pts)
    # CalculateOther over cFrame/dynaCam<i>.bmp: one point cloud per dynamic frame
    U = oracle.pipeline(spec, ph, gr, want=("U",))["U"]
    sw0, sb0 = oracle.strip_regression(dyn[0])
    for f in (1, 2, 3):
        sw1, sb1 = oracle.strip_regression(dyn[f])
        U = U + oracle.delta_p(sw0, sb0, sw1, sb1).astype(np.float64)
        pts = oracle.point_cloud(spec, oracle.triangulate(spec, U)["z"])
        assert open(os.path.join(d, "PointCloud", "cFrame%d.txt" % f)).read() == "".join("%g %g %g\n" % tuple(p) for p in pts), f
        sw0, sb0 = sw1, sb1


# ------------------------------------------------------------------ round 2: batch outputs in place, native gather, stream order
def test_decode_batch_ex_writes_row_tiles_into_a_full_height_map(api, oracle, synth, shard, torch_cuda):
    """slx_decode_batch_ex with a plane stride: every rank's row tile decoded straight into its rows of [set][H][W] (what the
    gather delivers), here all "ranks" in one process; ragged tile heights; depth plus the optional planes."""
    torch = torch_cuda
    spec = small_spec(synth, "C2", 128, 50)
    n_sets, H, W = 3, spec["height"], spec["width"]
    sets = [synth.random_planes(spec, seed=300 + s)[0] for s in range(n_sets)]
    ref = [oracle.pipeline(spec, p, None, want=("z", "x", "y", "U", "k", "mask")) for p in sets]
    full = {n: torch.full((n_sets, H, W), -7.0, dtype=torch.float64, device="cuda") for n in ("z", "x", "y", "U")}
    kfull = torch.full((n_sets, 2, H, W), -7, dtype=torch.int32, device="cuda")
    mfull = torch.full((n_sets, H, W), 9, dtype=torch.uint8, device="cuda")
    world = 3
    for rank in range(world):
        tile, lo, hi = shard.row_tile_spec(spec, world, rank)
        ph = torch.from_numpy(np.stack([p[:, lo:hi] for p in sets])).cuda()
        torch.cuda.synchronize()
        with api.Context(tile) as ctx:
            ctx.decode_batch_ex(n_sets, ph, None, z=full["z"][0, lo:], x=full["x"][0, lo:], y=full["y"][0, lo:], U=full["U"][0, lo:],
                                k=kfull[0, 0, lo:], mask=mfull[0, lo:], plane_stride=H * W)
            ctx.synchronize()
    for s in range(n_sets):
        for n in ("z", "x", "y", "U"):
            assert np.array_equal(full[n][s].cpu().numpy(), ref[s][n], equal_nan=True), (s, n)
        assert np.array_equal(kfull[s].cpu().numpy(), ref[s]["k"]), s
        assert np.array_equal(mfull[s].cpu().numpy(), ref[s]["mask"]), s


def test_north_star_row_tiles_at_full_size_against_the_oracle(api, oracle, synth, shard, torch_cuda):
    """BASELINE configuration 4 as north_star cuts it, at its real size: the eight 1920 x 150 row tiles (row_offset 0, 150 ... 1050)
    of 1920 x 1200 frame-sets.  (a) one frame-set, tile by tile through the single-set call: z and y of the eight tiles concatenate
    to the oracle's full maps; (b) a batch of 4 frame-sets, every "rank" decoding its tile of all of them in ONE launch of
    slx_decode_batch_ex straight into [set][1200][1920] (plane_stride = 1200 * 1920, what the gather's destination rank does):
    the assembled array equals the oracle's, set by set.  Tolerance: bit-equal (contract: 1e-4 mm RMS)."""
    torch = torch_cuda
    spec = synth.make_spec("C4")
    H, W, world = spec["height"], spec["width"], 8
    assert (W, H) == (1920, 1200)
    scenes = [synth.render(spec, "sphere", seed=91, noise_sigma=2.0)[0]] + [synth.random_planes(spec, seed=92 + s)[0] for s in range(3)]
    refs = [oracle.pipeline(spec, ph, None, want=("z", "y"), threads=8) for ph in scenes]
    # (a) the tiles of frame-set 0, one call each
    parts = []
    for rank in range(world):
        tile, lo, hi = shard.row_tile_spec(spec, world, rank)
        assert (tile["height"], tile["row_offset"], lo, hi) == (150, 150 * rank, 150 * rank, 150 * rank + 150)
        info = {}
        parts.append(api.decode_frameset(tile, scenes[0][:, lo:hi], None, want=("z", "y"), info=info))
        assert info["kernel"].startswith("slx_strip_kernel<3, 3, 0, 4, true>:"), info
    for w in ("z", "y"):
        assert np.array_equal(np.concatenate([p[w] for p in parts]), refs[0][w], equal_nan=True), w
    # (b) 4 frame-sets, each rank's tile of all of them in one launch, in place in the full-height maps
    n_sets = len(scenes)
    full = {w: torch.full((n_sets, H, W), -7.0, dtype=torch.float64, device="cuda") for w in ("z", "y")}
    for rank in range(world):
        tile, lo, hi = shard.row_tile_spec(spec, world, rank)
        ph = torch.from_numpy(np.stack([sc[:, lo:hi] for sc in scenes])).cuda()
        torch.cuda.synchronize()              # the context's stream does not order itself against torch's stream
        with api.Context(tile) as ctx:
            ctx.decode_batch_ex(n_sets, ph, None, z=full["z"][0, lo:], y=full["y"][0, lo:], plane_stride=H * W)
            ctx.synchronize()
            assert ctx.last_kernel().startswith("slx_strip_kernel<3, 3, 0, 4, true>:"), ctx.last_kernel()
    for s in range(n_sets):
        for w in ("z", "y"):
            assert np.array_equal(full[w][s].cpu().numpy(), refs[s][w], equal_nan=True), (s, w)


def test_config_5_row_tiles_at_full_size_against_the_oracle(api, oracle, synth, shard, torch_cuda):
    """BASELINE configuration 5 as north_star cuts it for 8 GPUs, at its real size: the eight 4096 x 375 row tiles (row_offset 0, 375 ...
    2625 -- the (v - cy) term of R/CCalculation.cpp:155-166 must keep referring to the full-frame row) of 4096 x 3000, 4-frequency x
    8-step frame-sets, on slx_strip_kernel<3, 4, 0, 8, *>.  (a) one frame-set, tile by tile: z and the fringe orders k of the eight
    tiles concatenate to the oracle's full maps; (b) two frame-sets, every "rank" decoding its tile of both in ONE launch of
    slx_decode_batch_ex straight into [set][3000][4096] (plane_stride = 3000 * 4096, what the gather's destination rank does).
    Tolerance: 1e-5 mm RMS in the contract; bit-equal here (phase indices exactly)."""
    torch = torch_cuda
    spec = synth.make_spec("C5")
    H, W, world = spec["height"], spec["width"], 8
    assert (W, H, spec["n_freq"], spec["n_steps"]) == (4096, 3000, 4, 8)
    scenes = [synth.render(spec, "tilted", seed=0x5EED + 55, noise_sigma=1.0)[0], synth.random_planes(spec, seed=56)[0]]
    refs = [oracle.pipeline(spec, ph, None, want=("z", "k"), threads=8) for ph in scenes]
    # (a) the tiles of frame-set 0, one call each, with the fringe orders beside the depth
    parts = []
    for rank in range(world):
        tile, lo, hi = shard.row_tile_spec(spec, world, rank)
        assert (tile["height"], tile["row_offset"], lo, hi) == (375, 375 * rank, 375 * rank, 375 * rank + 375)
        info = {}
        parts.append(api.decode_frameset(tile, scenes[0][:, lo:hi], None, want=("z", "k"), info=info))
        assert info["kernel"].startswith("slx_strip_kernel<3, 4, 0, 8, true>:"), info
    assert_same({w: np.concatenate([p[w] for p in parts], axis=-2) for w in ("z", "k")}, refs[0], ("z", "k"), RMS_TOL_MM_C5)
    for w in ("z", "k"):
        assert np.array_equal(np.concatenate([p[w] for p in parts], axis=-2), refs[0][w], equal_nan=True), w
    # (b) both frame-sets, each rank's tile of both in one launch, in place in the full-height maps; depth only: the bench's launch
    n_sets = len(scenes)
    full = torch.full((n_sets, H, W), -7.0, dtype=torch.float64, device="cuda")
    for rank in range(world):
        tile, lo, hi = shard.row_tile_spec(spec, world, rank)
        ph = torch.from_numpy(np.stack([sc[:, lo:hi] for sc in scenes])).cuda()
        torch.cuda.synchronize()              # the context's stream does not order itself against torch's stream
        with api.Context(tile) as ctx:
            ctx.decode_batch_ex(n_sets, ph, None, z=full[0, lo:], plane_stride=H * W)
            ctx.synchronize()
            assert ctx.last_kernel().startswith("slx_strip_kernel<3, 4, 0, 8, false>:"), ctx.last_kernel()
    for s_ in range(n_sets):
        got = full[s_].cpu().numpy()
        assert_same({"z": got}, refs[s_], ("z",), RMS_TOL_MM_C5)
        assert np.array_equal(got, refs[s_]["z"], equal_nan=True), s_


@pytest.mark.parametrize("name,shape,world,n_sets,chunk", [("C4", (128, 37), 3, 5, 2), ("C2", (64, 41), 8, 4, 3), ("C4", (1920, 1200), 8, 8, 8)])
def test_staged_gather_row_scatter_on_one_gpu(api, oracle, synth, shard, torch_cuda, name, shape, world, n_sets, chunk):
    """The STAGED gather shape with every rank played on this one GPU: each "rank" decodes its row tile of all frame-sets into a dense
    tile stack (what slx_decode_gather's scratch is), the root decodes in place; per chunk the messages libslx plans
    (slx_gather_plan_ex, every rank's plan matched pairwise as RCCL would) are carried by device-to-device copies into the root's
    staging slot, and slx_scatter_rows -- the kernel the RCCL path launches on the root -- moves the tiles to their rows.  The
    assembled [set][H][W] must be the oracle's, bit for bit; the last case is BASELINE configuration 4's real geometry (8 ranks,
    150-row tiles of 1920 x 1200, chunks of 8: 7 messages of 18.4 MB into a 129 MB slot).  RCCL itself needs N GPUs; this pins
    everything around it."""
    torch = torch_cuda
    spec = small_spec(synth, name, *shape)
    H, W = spec["height"], spec["width"]
    sets = [synth.random_planes(spec, seed=880 + s)[0] for s in range(n_sets)]
    want = np.stack([oracle.pipeline(spec, p, None, want=("z",), threads=8)["z"] for p in sets])
    table = shard.shards_by_rows(n_sets, world, H)
    full = torch.full((n_sets, H, W), -7.0, dtype=torch.float64, device="cuda")
    flat_full = full.view(-1)
    local = {}
    for rank in range(world):
        tile, lo, hi = shard.row_tile_spec(spec, world, rank)
        ph = torch.from_numpy(np.stack([p[:, lo:hi] for p in sets])).cuda()
        torch.cuda.synchronize()
        with api.Context(tile) as ctx:
            if rank == 0:
                ctx.decode_batch_ex(n_sets, ph, None, z=full[0, lo:], plane_stride=H * W)       # the root: in place
            else:
                local[rank] = torch.empty((n_sets, hi - lo, W), dtype=torch.float64, device="cuda")
                ctx.decode_batch(n_sets, ph, None, local[rank])
            ctx.synchronize()
    n_msgs = 0
    with api.Context(spec) as ctx:
        for first in range(0, n_sets, chunk):
            root_msgs, scat, staging = api.gather_plan_ex(table, 0, H, W, first, chunk, local_plane_stride=H * W, root=0, shape="staged")
            assert all(kind == 2 for _, kind, _, _ in root_msgs) and len(scat) == len(root_msgs) == sum(1 for t in table[1:] if t[3])
            stage = torch.full((staging,), -3.0, dtype=torch.float64, device="cuda")
            for peer, _, roff, rcnt in root_msgs:
                sends, no_scatter, no_staging = api.gather_plan_ex(table, peer, H, W, first, chunk, root=0, shape="staged")
                assert len(sends) == 1 and not no_scatter and no_staging == 0
                to, kind, soff, scnt = sends[0]
                assert (to, kind, scnt) == (0, 1, rcnt)
                stage[roff:roff + rcnt] = local[peer].view(-1)[soff:soff + scnt]
                n_msgs += 1
            torch.cuda.synchronize()
            ctx.scatter_rows(scat, stage, flat_full)
            ctx.synchronize()
    assert n_msgs == (world - 1 - sum(1 for t in table[1:] if not t[3])) * ((n_sets + chunk - 1) // chunk)
    assert np.array_equal(full.cpu().numpy(), want, equal_nan=True)


@pytest.mark.parametrize("split", ["framesets", "rows"])
@pytest.mark.parametrize("root", [0, -1])
def test_native_gather_world_of_one(api, oracle, synth, shard, torch_cuda, split, root):
    """slx_comm_* / slx_gather_depth / slx_decode_gather on RCCL with the one rank this box has: communicator creation from a
    unique id, the in-place path, the copy path for a separate local buffer, chunking with a ragged last chunk.  (N > 1 message
    patterns: tests/test_shard_gloo.py runs the same shard tables through torch.distributed.)"""
    torch = torch_cuda
    spec = small_spec(synth, "C4", 128, 40)
    n_sets, H, W = 5, spec["height"], spec["width"]
    sets = [synth.random_planes(spec, seed=400 + s)[0] for s in range(n_sets)]
    want = np.stack([oracle.pipeline(spec, p, None, want=("z",))["z"] for p in sets])
    table = (shard.shards_by_frameset if split == "framesets" else shard.shards_by_rows)(n_sets, 1, H)
    phase = torch.from_numpy(np.stack(sets)).cuda()
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        comm = api.Comm(ctx, api.comm_unique_id(), 1, 0)
        assert (comm.world, comm.rank) == (1, 0)
        full = torch.full((n_sets, H, W), -1.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        comm.decode_gather(table, H, 2, phase, None, None, full, root=root)        # chunks of 2, 2, 1 frame-sets
        comm.synchronize()
        assert np.array_equal(full.cpu().numpy(), want, equal_nan=True)
        # a separate local buffer: the gather copies this rank's own shard into place
        local = torch.empty((n_sets, H, W), dtype=torch.float64, device="cuda")
        full.fill_(-1.0)
        torch.cuda.synchronize()
        ctx.decode_batch(n_sets, phase, None, local)
        ctx.synchronize()
        comm.gather_depth(table, H, W, local, full, root=root)
        comm.synchronize()
        assert np.array_equal(full.cpu().numpy(), want, equal_nan=True)
        with pytest.raises(api.SlxError):
            comm.gather_depth([(0, n_sets, 0, H + 1)], H, W, local, full)          # a shard taller than the frame
        # the communicator belongs to the device, not to one context: a second context (another tuning, as bench.py's second
        # split would be another tile) decodes through it too
        with api.Context(spec) as other:
            other.set_tuning(strip_rows=4)
            full.fill_(-1.0)
            torch.cuda.synchronize()
            comm.decode_gather(table, H, 3, phase, None, None, full, root=root, ctx=other)
            comm.synchronize()
            assert np.array_equal(full.cpu().numpy(), want, equal_nan=True)
        comm.close()


def test_host_frames_can_be_replaced_right_after_decode(api, oracle, synth):
    """slx_decode is asynchronous; slx_set_frame(SLX_MEM_HOST) rewrites the staging buffers the decode reads.  The library
    orders the two (it waits for the event of the last launch), so decode(); set_frame(next) keeps the first result intact."""
    spec = synth.make_spec("C3")                                       # 24 planes of 1920 x 1200: the decode takes a while
    a = synth.random_planes(spec, seed=1)
    b = synth.random_planes(spec, seed=2)
    ra = oracle.pipeline(spec, a[0], a[1], want=("z",), threads=8)["z"]
    rb = oracle.pipeline(spec, b[0], b[1], want=("z",), threads=8)["z"]
    with api.Context(spec) as ctx:
        ctx.set_frames(*a)
        for _ in range(3):
            ctx.decode()
            ctx.set_frames(*b)                                         # at once, no synchronize in between
            za = ctx.get_depth()
            ctx.decode()
            ctx.set_frames(*a)
            zb = ctx.get_depth()
            assert np.array_equal(za, ra, equal_nan=True) and np.array_equal(zb, rb, equal_nan=True)


def test_pipe_and_a_second_context_run_concurrently(api, oracle, synth, torch_cuda):
    """Waiting for one context's result must not depend on, or disturb, other work on the device: an ingest pipe keeps its slots
    full while a second context decodes on a caller stream and reads its outputs back; both stay bit-equal to the oracle."""
    torch = torch_cuda
    spec = small_spec(synth, "C2", 256, 120)
    sets = [synth.random_planes(spec, seed=500 + i)[0] for i in range(8)]
    want = [oracle.pipeline(spec, p, None, want=("z",))["z"] for p in sets]
    other_spec = small_spec(synth, "C1x4", 320, 100)
    oph, ogr = synth.random_planes(other_spec, seed=77)
    owant = oracle.pipeline(other_spec, oph, ogr, want=("z", "U"))
    s = torch.cuda.Stream()
    with api.Context(spec) as pctx, api.Context(other_spec, aux=("U",)) as octx:
        pipe = api.Pipe(pctx, slots=3, sets_per_slot=1)
        octx.set_frames(torch.from_numpy(oph).cuda(), torch.from_numpy(ogr).cuda())
        torch.cuda.synchronize()
        got, in_flight = [], 0
        for i, p in enumerate(sets):
            if in_flight == 3:
                got.append(np.array(pipe.collect()[0]))
                in_flight -= 1
            buf = pipe.acquire()
            buf[0, :, :, :spec["width"]] = p
            pipe.submit(1)
            in_flight += 1
            octx.decode(stream=s.cuda_stream)                          # on a caller stream, between the pipe's submissions
            for w in ("z", "U"):
                assert np.array_equal(octx.get_output(w), owant[w], equal_nan=True), (i, w)
        while in_flight:
            got.append(np.array(pipe.collect()[0]))
            in_flight -= 1
        pipe.close()
    for i in range(8):
        assert np.array_equal(got[i], want[i], equal_nan=True), i


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2` as typed: the parent starts the ranks.  With one GPU on the box the ranks share it and talk
    over gloo (--rehearse-on-one-gpu); the line must be valid and carry both gather splits, checked against the local decodes."""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "20", "--warmup", "5",
                        "--sets-per-gpu", "4"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    # the rehearsal talks over gloo: no RCCL communicator exists, and the line says so instead of repeating torch's count
    assert d["n_gpus"] == 2 and d["torch_world_size"] == 2 and d["rccl_world_size"] is None and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert "root's ingest" in d["config"]["sharding"] and "kernel_only" in d["config"]["sharding"]
    # N > 1: the headline is north_star's split end to end (row tiles + the gather), exactly --steps steps; the decode alone sits beside it
    # ... in the faster of the two gather shapes of that split (both timed for --steps steps, both checked against the local decodes)
    best = d["with_gather"]["value_is"]
    assert best in ("rows", "rows_staged") and d["config"]["gather_shape"] == d["with_gather"][best]["gather_shape"]
    assert d["value"] == d["with_gather"][best]["end_to_end"]["value"] == max(d["with_gather"][k]["end_to_end"]["value"] for k in ("rows", "rows_staged"))
    assert d["with_gather"]["rows"]["steps"] == d["with_gather"]["rows_staged"]["steps"] == 20 and d["stuck"] is None
    assert d["kernel_only"]["value"] > d["value"] and "row-tiled" in d["config"]["workload"]
    for key, shape, msgs in (("framesets", "in_place", 1), ("rows", "in_place", 8), ("rows_staged", "staged", 1)):
        g = d["with_gather"][key]
        assert "error" not in g, g
        assert g["gather_shape"] == shape and g["gathered_equals_local_decodes"] is True and g["gathered_shape"] == [8, 1200, 1920]
        assert g["kernel_only"]["value"] > 0 and g["end_to_end"]["value"] > 0 and g["gather_only"]["ms_per_step"] > 0
        assert g["messages_at_root_per_step"] == msgs, g                # 2 ranks, 8 frame-sets, one chunk: per (peer, set) / per (peer, chunk)
        assert g["gathered_equals_oracle"] is True, g                   # rank 0: frame-set 0 and one of the last chunk against the oracle
    # N > 1: the line certifies itself -- gathered maps (every rank's tile in them) against the oracle, not only against the ranks' own decodes
    assert d["parity_vs_oracle"] is True
    checks = d["with_gather"]["oracle_checks"]
    assert len(checks) == 6 and all(c["equal"] is True for c in checks) and {c["measurement"] for c in checks} == {"rows", "rows_staged", "framesets"}
    assert {c["gathered_set"] for c in checks} == {0, 4}               # 8 gathered sets in one chunk of 8: set 0 and the middle of the last chunk
    assert d["with_gather"]["rows_staged"]["root_staging_bytes"] == 2 * 8 * 600 * 1920 * 8 and d["with_gather"]["rows"]["root_staging_bytes"] == 0


def test_bench_gather_code_path_on_rccl_with_one_rank():
    """`bench.py --gather-world-of-one`: the N > 1 code path of the bench ON RCCL with the one rank this box has -- the process group,
    the communicator from a unique id, slx_comm_set_gather_shape, slx_decode_gather in chunks, the gather alone, the checksums, for
    both gather shapes and the frame-set split -- so that the first 8-GPU run does not execute a line of Python or C that never ran.
    (With one rank nothing travels: the messages themselves are covered by the plan replays and the one-GPU staged test.)"""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gather-world-of-one", "--steps", "20", "--warmup", "5", "--sets-per-gpu", "8",
                        "--no-cpu-baseline", "--no-other-configs", "--no-traffic-probe", "--no-power-probe"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_world_size"] == 1 and d["rccl_rank_of_rank0"] == 0 and d["stuck"] is None
    assert d["value"] == d["kernel_only"]["value"] > 0                 # nobody to gather from: the line's value stays the decode's
    for key, shape in (("rows", "in_place"), ("rows_staged", "staged"), ("framesets", "in_place")):
        g = d["with_gather"][key]
        assert "error" not in g, g
        assert g["gather_shape"] == shape and g["gathered_equals_local_decodes"] is True and g["gathered_shape"] == [8, 1200, 1920]
        assert g["messages_at_root_per_step"] == 0 and g["bytes_into_root_per_step"] == 0
        assert g["gathered_equals_oracle"] is True, g
        assert g["end_to_end"]["value"] > 0 and g["kernel_only"]["value"] > 0 and g["gather_only"]["ms_per_step"] >= 0
    assert d["parity_vs_oracle"] is True and len(d["with_gather"]["oracle_checks"]) == 6


@pytest.mark.parametrize("bits", [1, 3, 6, 7, 8, 10, 12])
@pytest.mark.parametrize("std_lut", [True, False])
def test_strip_kernel_gray_widths(api, oracle, synth, bits, std_lut):
    """The strip kernel packs the Gray bits of a quad's four pixels in one register (up to 8 bits per pixel; wider codes go
    pixel by pixel): every width class, the reflected-code shortcut and an arbitrary table, exact ties included, in the
    reference's mode and the Gray-mask mode."""
    for name in ("C1x4", "C3"):
        spec = small_spec(synth, name, 256, 40)
        spec["gray_bits"] = bits
        spec["gray_stripe"] = max(1, spec["proj_width"] // (1 << bits))
        if name == "C1x4":
            spec["periods"] = [max(2, 2 * spec["gray_stripe"])]
        rng = np.random.default_rng(bits)
        spec["gray_lut"] = synth.standard_gray_lut(bits) if std_lut else rng.integers(-50, 3000, 1 << bits).astype(np.int16)
        ph, gr = synth.random_planes(spec, seed=900 + bits)
        gr[:, :, :128] = np.where(gr[:, :, :128] > 127, 220, 20)
        gr[1::2, :, 64:96] = gr[0::2, :, 64:96]                        # exact ties: pattern == inverse -> bit 0
        ref = oracle.pipeline(spec, ph, gr, want=("z",))
        for variant in (api.VARIANT_STRIP, api.VARIANT_GENERIC):
            got = api.decode_frameset(spec, ph, gr, want=("z",), variant=variant)
            assert_same(got, ref, ("z",))


@pytest.mark.parametrize("name", ["C1", "C1x4", "C2", "C3", "C4", "C5", "C5x4"])
@pytest.mark.parametrize("shape", [(7, 64), (33, 1024), (67, 500), (1, 4), (130, 256), (37, 1920)])
def test_strip_kernel_optional_planes(api, oracle, synth, torch_cuda, name, shape):
    """x, y, U, k and the mask from the strip kernel: every mode and step count it runs, ragged geometries, unstructured bytes
    (so every branch of the unwrap / merge / mask is taken), one frame-set through slx_decode and a batch of three through
    slx_decode_batch_ex into planes with a pitch (as a row tile lands in a full-height map); any subset of the planes."""
    torch = torch_cuda
    h, w = shape
    spec = small_spec(synth, "C5" if name == "C5x4" else name, w, h)
    if name == "C5x4":
        spec["n_steps"] = 4
    want = strip_outputs(spec)
    sets = [synth.random_planes(spec, seed=h * 31 + w + i) for i in range(3)]
    if sets[0][1] is not None and w >= 8:
        for _, gr in sets:
            gr[:, :, : w // 2] = np.where(gr[:, :, : w // 2] > 127, 220, 20)
    refs = [oracle.pipeline(spec, ph, gr, want=want) for ph, gr in sets]
    got = api.decode_frameset(spec, sets[0][0], sets[0][1], want=want, variant=api.VARIANT_STRIP)
    assert_same(got, refs[0], want)
    sub = tuple(x for x in want if x in ("z", "y", "mask"))
    got = api.decode_frameset(spec, sets[0][0], sets[0][1], want=sub, variant=api.VARIANT_STRIP)
    assert_same(got, refs[0], sub)
    # batch, planes 2 rows taller than the tile, the tile starting at row 1
    Hp = h + 2
    phase = torch.from_numpy(np.stack([ph for ph, _ in sets])).cuda()
    gray = torch.from_numpy(np.stack([gr for _, gr in sets])).cuda() if sets[0][1] is not None else None
    nk = spec["n_freq"] - 1 if (spec["mode"] in (3, 4) and spec["n_freq"] > 1) else 0
    out = {n: torch.full((3, Hp, w), -5.0, dtype=torch.float64, device="cuda") for n in ("z", "x", "y", "U")}
    kk = torch.full((3, max(nk, 1), Hp, w), -5, dtype=torch.int32, device="cuda")
    mm = torch.full((3, Hp, w), 7, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        ctx.set_variant(api.VARIANT_STRIP)
        ctx.decode_batch_ex(3, phase, gray, z=out["z"][0, 1:], x=out["x"][0, 1:], y=out["y"][0, 1:], U=out["U"][0, 1:],
                            k=kk[0, 0, 1:] if nk else None, mask=mm[0, 1:], plane_stride=Hp * w)
        ctx.synchronize()
    for i in range(3):
        for n in ("z", "x", "y", "U"):
            a = out[n][i].cpu().numpy()
            assert np.array_equal(a[1:h + 1], refs[i][n], equal_nan=True), (i, n)
            assert np.all(a[0] == -5.0) and np.all(a[h + 1] == -5.0), (i, n)             # nothing outside the tile's rows
        if nk:
            a = kk[i].cpu().numpy()
            assert np.array_equal(a[:, 1:h + 1], refs[i]["k"]) and np.all(a[:, 0] == -5) and np.all(a[:, h + 1] == -5), i
        a = mm[i].cpu().numpy()
        want_mask = refs[i]["mask"] if "mask" in refs[i] else np.ones((h, w), np.uint8)
        assert np.array_equal(a[1:h + 1], want_mask) and np.all(a[0] == 7) and np.all(a[h + 1] == 7), i


@pytest.mark.parametrize("split", ["rows", "framesets"])
def test_cpp_gather_host_loop(tmp_path, oracle, synth, split):
    """tests/cpp/gather_host_loop.cpp: a C++ rank of the multi-GPU host loop through the C ABI alone -- communicator from a
    unique id handed over in a file, slx_decode_gather in chunks, slx_gather_depth from a separate buffer -- as a world of one
    (this box has one GPU; the program takes rank / world arguments for a node)."""
    import subprocess
    from conftest import ROOT, _ensure_built
    _ensure_built()
    exe = os.path.join(ROOT, "tests", "cpp", "gather_host_loop")
    assert os.path.exists(exe)
    W, H, sets = 128, 37, 5
    spec = {"name": "gather", "width": W, "height": H, "row_offset": 0, "proj_width": 1920, "mode": synth.MODE_MULTIFREQ, "n_freq": 3, "n_steps": 4,
            "periods": [1920, 240, 30], "gray_bits": 0, "gray_stripe": 0, "gray_lut": None, "fov_min": -1e300, "fov_max": 1e300,
            "calib": {"cam": [3600, 0, (W - 1) / 2.0, 0, 3600, (H - 1) / 2.0, 0, 0, 1], "pro": [3000, 0, 900, 0, 3000, 600, 0, 0, 1],
                      "rot": [0.99, -0.01, 0.13, 0.02, 0.99, -0.1, -0.13, 0.1, 0.98], "trans": [-31.7, -9.3, 39.4]}}
    planes = [synth.random_planes(spec, seed=700 + s)[0] for s in range(sets)]
    np.stack(planes).tofile(str(tmp_path / "in.bin"))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([exe, "0", "1", str(tmp_path / "id.bin"), split, str(W), str(H), str(sets), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    got = np.fromfile(str(tmp_path / "out.bin"), dtype=np.float64).reshape(sets, H, W)
    for s in range(sets):
        assert np.array_equal(got[s], oracle.pipeline(spec, planes[s], None, want=("z",))["z"], equal_nan=True), s


@pytest.mark.parametrize("world,split,shape,size,sets,chunk", [(3, "rows", "staged", (128, 37), 5, 2), (3, "rows", "in_place", (128, 37), 5, 2),
                                                                (8, "rows", "staged", (256, 50), 7, 3), (8, "rows", "in_place", (256, 50), 7, 3),
                                                                (4, "framesets", "staged", (64, 20), 9, 2), (2, "rows", "staged", (1920, 1200), 6, 2)])
def test_gather_code_with_peers_as_threads_of_one_process(tmp_path, oracle, synth, world, split, shape, size, sets, chunk):
    """tests/cpp/gather_threads.cpp: every rank of the multi-GPU host loop as a THREAD on this one GPU, libslx.so as built, only the
    wire replaced (the executable defines the nccl* entry points: grouped sends / receives matched per ordered pair of ranks in
    posting order, equal counts demanded, data moved in stream order by device copies).  So the library's gather code runs WITH
    PEERS: gather_range, the two staging slots of the staged shape and the events that order their reuse, the row-scatter kernel on
    its own stream, the chunk pipeline of slx_decode_gather (twice in a row: the second call must wait for the first one's gather),
    slx_gather_depth from separate local buffers.  Worlds of 2 - 8, ragged tiles, ragged last chunks, both shapes, both splits, and
    configuration 4's real frame size; the gathered maps against the oracle, bit for bit.  (What this cannot show is RCCL and xGMI
    themselves: that is the driver's 8-GPU run.)"""
    import subprocess
    from conftest import ROOT, _ensure_built
    _ensure_built()
    exe = os.path.join(ROOT, "tests", "cpp", "gather_threads")
    assert os.path.exists(exe)
    W, H = size
    spec = {"name": "gather", "width": W, "height": H, "row_offset": 0, "proj_width": 1920, "mode": synth.MODE_MULTIFREQ, "n_freq": 3, "n_steps": 4,
            "periods": [1920, 240, 30], "gray_bits": 0, "gray_stripe": 0, "gray_lut": None, "fov_min": -1e300, "fov_max": 1e300,
            "calib": {"cam": [3600, 0, (W - 1) / 2.0, 0, 3600, (H - 1) / 2.0, 0, 0, 1], "pro": [3000, 0, 900, 0, 3000, 600, 0, 0, 1],
                      "rot": [0.99, -0.01, 0.13, 0.02, 0.99, -0.1, -0.13, 0.1, 0.98], "trans": [-31.7, -9.3, 39.4]}}
    planes = [synth.random_planes(spec, seed=1700 + s)[0] for s in range(sets)]
    np.stack(planes).tofile(str(tmp_path / "in.bin"))
    r = subprocess.run([exe, str(world), split, shape, str(W), str(H), str(sets), str(chunk), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    got = np.fromfile(str(tmp_path / "out.bin"), dtype=np.float64).reshape(sets, H, W)
    for s in range(sets):
        assert np.array_equal(got[s], oracle.pipeline(spec, planes[s], None, want=("z",), threads=8)["z"], equal_nan=True), s


# ------------------------------------------------------------------ multi-frequency: every input of a coarser frequency
@pytest.mark.parametrize("periods", [[1920, 240, 30], [4096, 512, 64, 8], [1000, 37], [16384, 9], [777, 333, 111]])
def test_coarse_frequencies_exhaustive_inputs(api, oracle, synth, periods):
    """The coarser frequencies only decide fringe orders; every one of the 511 x 511 inputs of a coarser frequency, for each
    coarser frequency in turn, against unstructured bytes in the others: depth through the strip kernel equals the oracle's."""
    F = len(periods)
    planes = exhaustive_planes(512)
    spec = dict(synth.make_spec("C4"), width=512, height=511, periods=periods, n_freq=F)
    spec["calib"] = synth.scaled_calibration(512, 511, 1920)
    spec["fov_min"], spec["fov_max"] = -1e300, 1e300
    rng = np.random.default_rng(F * 1000 + periods[0])
    for sweep in range(F - 1):
        ph = rng.integers(0, 256, size=(F * 4, 511, 512), dtype=np.uint8)
        ph[sweep * 4:sweep * 4 + 4] = planes
        ref = oracle.pipeline(spec, ph, None, want=("z",), threads=8)["z"]
        got = api.decode_frameset(spec, ph, None, want=("z",), variant=api.VARIANT_STRIP)["z"]
        assert np.array_equal(got, ref, equal_nan=True), (periods, sweep, int((got != ref).sum()))


def test_fringe_order_ties_and_wrap_edges(api, oracle, synth, torch_cuda):
    """Inputs that sit ON the decisions of the temporal unwrap: exact ties (quarter-turn phases) and coarse phases at the wrap
    of pix (angles just below 360 degrees and at 0), mixed with unstructured bytes, as a batch."""
    torch = torch_cuda
    spec = small_spec(synth, "C4", 256, 64)
    spec["fov_min"], spec["fov_max"] = -1e300, 1e300
    sets = [quadrant_tie_planes(spec)]
    rng = np.random.default_rng(5)
    for amp in (255, 200, 3):
        ph = rng.integers(0, 256, size=(12, 64, 256), dtype=np.uint8)
        for f in (0, 1):                                               # sine term -1, 0, +1 against a large cosine term: 360-, 0, 0+
            s = rng.integers(-1, 2, size=(64, 256))
            ph[f * 4 + 0] = 128 + np.maximum(s, 0)
            ph[f * 4 + 2] = 128 + np.maximum(-s, 0)
            ph[f * 4 + 1] = amp
            ph[f * 4 + 3] = 0
        sets.append(ph)
    ref = [oracle.pipeline(spec, ph, None, want=("z",))["z"] for ph in sets]
    phase = torch.from_numpy(np.stack(sets)).cuda()
    z = torch.empty((len(sets), 64, 256), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with api.Context(spec) as ctx:
        ctx.set_variant(api.VARIANT_STRIP)
        ctx.decode_batch(len(sets), phase, None, z)
        ctx.synchronize()
    got = z.cpu().numpy()
    for i in range(len(sets)):
        assert np.array_equal(got[i], ref[i], equal_nan=True), (i, int((got[i] != ref[i]).sum()))
