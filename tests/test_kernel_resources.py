"""Register / scratch budget of every strip-kernel instantiation, read from the compiled gfx950 code (CPU only: hipcc cross-compiles).

The DMA ring of `slx_strip_kernel` (csrc/slx_kernels.hip) waits with counted `s_waitcnt vmcnt(n)`: the counts name the vector-memory
operations the source issues per step.  They are exact only while hipcc adds none of its own -- a spilled VGPR is a scratch store and
a scratch load -- so a spill would turn the waits into reads of LDS slots whose DMA has not landed, something only the GPU box could
catch.  This test turns that invariant into a build-time fact: no instantiation spills a vector register or owns a private segment.
It also pins the occupancy the launch planner assumes (slx_strip_waves_per_simd, csrc/slx_plan.cpp) to the compiled VGPR counts.
"""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "structured-light-calculation_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

MODE_GRAY_PHASE, MODE_MULTIFREQ, MODE_MULTIFREQ_GRAYMASK = 2, 3, 4


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    """{(mode, F, GB, NS, AUX): metadata dict} of every slx_strip_kernel instantiation in the device code."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is not installed")
    out = str(tmp_path_factory.mktemp("asm") / "slx_kernels.s")
    # the flags of csrc/Makefile
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, "slx_kernels.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    found = {}
    for blk in re.split(r"\n  - \.agpr_count:", meta):
        m = re.search(r"\.name:\s+\S*slx_strip_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)E", blk)
        if not m:
            continue
        key = tuple(int(g) for g in m.groups())
        found[key] = {f: int(re.search(r"\.%s:\s+(\d+)" % f, blk).group(1))
                      for f in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    assert len(found) >= 36, "expected every strip-kernel instantiation in the assembly, found %d" % len(found)
    # the decoder objects' kernels (modes 0 and 1): keyed like the others, with the counts their launch plan uses
    for blk in re.split(r"\n  - \.agpr_count:", meta):
        m = re.search(r"\.name:\s+\S*slx_decoder_strip_kernelILi(\d)E", blk)
        if m:
            mode = int(m.group(1))
            found[(mode, 1 if mode == 0 else 0, 0, 4, 0)] = {f: int(re.search(r"\.%s:\s+(\d+)" % f, blk).group(1))
                                                             for f in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                                                                       "group_segment_fixed_size")}
    assert (0, 1, 0, 4, 0) in found and (1, 0, 0, 4, 0) in found
    # the stream kernel (resident waves, queues): mode 3, 4 steps, keyed with GB = 9 to keep it apart from the strip instantiations
    for blk in re.split(r"\n  - \.agpr_count:", meta):
        m = re.search(r"\.name:\s+\S*slx_stream_kernelILi(\d)E", blk)
        if m:
            found[(3, int(m.group(1)), 9, 4, 0)] = {f: int(re.search(r"\.%s:\s+(\d+)" % f, blk).group(1))
                                                                  for f in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                                                                            "group_segment_fixed_size")}
    assert all((3, F, 9, 4, 0) in found for F in (1, 2, 3, 4))
    # the stream kernel of the reference's own mode (round 6): mode 2, one frequency, keyed with GB = 8
    for blk in re.split(r"\n  - \.agpr_count:", meta):
        if re.search(r"\.name:\s+\S*slx_gstream_kernelE", blk):
            found[(2, 1, 8, 4, 0)] = {f: int(re.search(r"\.%s:\s+(\d+)" % f, blk).group(1))
                                      for f in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    assert (2, 1, 8, 4, 0) in found
    return found


def test_no_strip_kernel_spills_vector_registers_or_uses_scratch(kernels):
    bad = {k: v for k, v in kernels.items() if v["vgpr_spill_count"] != 0 or v["private_segment_fixed_size"] != 0}
    assert not bad, "hipcc added vector-memory operations of its own (the counted vmcnt waits of the DMA ring are no longer exact): %r" % bad


def test_strip_kernels_use_only_dynamic_lds(kernels):
    # the launch planner sizes the LDS (ring + staging per wave); a static allocation would not be in its arithmetic
    assert all(v["group_segment_fixed_size"] == 0 for v in kernels.values())


def test_kernels_without_optional_planes_fit_four_waves_per_simd(kernels):
    """512 VGPRs per SIMD lane: 4 waves need <= 128 each.  Every instantiation the headline configurations use (no x / y / U / k planes)
    must stay there: the LDS ring is sized for 16 waves per CU."""
    over = {k: v["vgpr_count"] for k, v in kernels.items() if k[4] == 0 and v["vgpr_count"] > (64 if k[0] in (0, 1) else 128)}
    assert not over, over


def test_planner_occupancy_matches_compiled_register_counts(kernels):
    """slx_strip_waves_per_simd (the planner's table) never claims more waves than floor(512 / VGPRs rounded up to 8) allows, and
    claims 4 wherever the compiled code allows 4."""
    lib_path = os.path.join(ROOT, "structured-light-calculation_amd", "libslx.so")
    if not os.path.exists(lib_path):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(lib_path)
    lib.slx_strip_waves_per_simd.restype = ctypes.c_uint
    lib.slx_strip_waves_per_simd.argtypes = [ctypes.c_int] * 5
    for (mode, F, GB, NS, AUX), v in sorted(kernels.items()):
        if mode == MODE_GRAY_PHASE and F != 1:
            continue                                   # instantiated by the template switch, never launched
        if GB in (8, 9):
            continue                                   # the stream kernels: planned for 4 waves per SIMD, checked by the <= 128 test above
        alloc = (v["vgpr_count"] + 7) // 8 * 8
        allowed = min(8, 512 // alloc)
        claimed = lib.slx_strip_waves_per_simd(mode, F, GB, NS, AUX)
        assert claimed <= allowed, ((mode, F, GB, NS, AUX), v["vgpr_count"], claimed, allowed)
        # the decode kernels are planned for 4 waves per SIMD (their LDS ring allows no more), the decoder kernels (modes 0, 1) for 8
        assert claimed == min(8 if mode in (0, 1) else 4, allowed), ((mode, F, GB, NS, AUX), v["vgpr_count"], claimed, allowed)


def _sgprs(operand_text):
    """Indices of every scalar register an operand list names (s5, s[4:7])."""
    regs = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", operand_text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", operand_text):
        regs.add(int(m.group(1)))
    return regs


def test_stream_kernel_ticket_register_is_untouched_between_issue_and_wait(tmp_path):
    """slx_stream_kernel takes its work tickets with a scalar atomic whose result lands in an SGPR when the NEXT `s_waitcnt lgkmcnt(0)`
    retires -- issue and wait are two asm statements with the step's DMA wait, the depth stores and the LDS reads between them
    (csrc/slx_kernels.hip: fetch_issue / fetch_wait).  The register allocator knows the variable is live across that window, so it
    gives the register to nothing else there; what it does NOT know is that the value is still in flight: a copy or a spill of it
    inside the window (a phi of the `if (ri == 0)` arms, scalar-register pressure) would read the placeholder (1) instead of the
    ticket, and rows would be skipped or decoded twice with no error.  This pins the compiled code of every instantiation:
      * every s_atomic_add writes a register R directly behind `s_mov_b32 R, 1` (two ticket variables: the entry ticket, the loop's);
      * no instruction READS R at a point an in-flight ticket can reach: "in flight" is set by the issue, cleared by an
        `s_waitcnt ... lgkmcnt(0)` or another definition of R, and propagated over the control-flow graph to a fixed point (the
        optional-plane instantiations are short of scalar registers and reuse R for other values outside the window: those uses
        are clear of it by construction of the walk)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is not installed")
    out = str(tmp_path / "slx_kernels.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, "slx_kernels.hip"), "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    ticket_reads = 0
    instantiations = [("slx_stream_kernelILi%dEEEv10SlxKParams" % F, "slx_stream_kernel<%d>" % F) for F in (1, 2, 3, 4)]
    instantiations += [("slx_gstream_kernelE10SlxKParams", "slx_gstream_kernel")]
    for mangled, who in instantiations:
        start = next(i for i, ln in enumerate(lines) if re.match(r"^_ZN\S*%s:" % mangled, ln))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = [ln.split(";")[0].rstrip() for ln in lines[start + 1:end]]
        # basic blocks: a label starts one, a branch / s_endpgm ends one
        blocks, label_of = [[]], {}
        for ln in body:
            m = re.match(r"^(\.LBB\S*):", ln)
            if m:
                if blocks[-1]:
                    blocks.append([])
                label_of[m.group(1)] = len(blocks) - 1
                continue
            if not ln.startswith("\t") or ln.strip().startswith("."):
                continue
            t = ln.strip()
            blocks[-1].append(t)
            if t.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                blocks.append([])
        succ = []
        for b, insts in enumerate(blocks):
            last = insts[-1] if insts else ""
            nxt = [b + 1] if b + 1 < len(blocks) else []
            if last.startswith("s_endpgm"):
                succ.append([])
            elif last.startswith("s_branch"):
                succ.append([label_of[last.split()[1]]])
            elif last.startswith("s_cbranch"):
                succ.append(nxt + [label_of[last.split()[1]]])
            else:
                assert not last.startswith(("s_setpc", "s_swappc")), (who, last)
                succ.append(nxt)
        atomics = [t for insts in blocks for t in insts if t.startswith("s_atomic_add ")]
        assert len(atomics) >= 2, (who, "expected the entry ticket and the loop's ticket")
        dest = sorted({int(re.match(r"s_atomic_add s(\d+),", t).group(1)) for t in atomics})
        assert len(dest) <= 2, (who, "more ticket registers than ticket variables (the entry ticket, the loop's): a copy of a pending register", dest)
        for R in dest:
            mov = "s_mov_b32 s%d, 1" % R

            def operands(t):
                op, _, rest = t.partition(" ")
                return op, [o.strip() for o in rest.split(",")] if rest else []

            def writes_R(t):
                op, ops = operands(t)
                if not ops or op.startswith(("s_cmp", "s_bitcmp", "s_waitcnt", "s_cbranch", "s_branch", "buffer_store", "global_store", "ds_write", "s_setprio", "s_nop")):
                    return False
                return R in _sgprs(ops[0])                               # ALU / scalar loads / the atomic: operand 0 is the destination

            def reads_R(t):
                op, ops = operands(t)
                if t.startswith("s_atomic_add s%d," % R):
                    return False                                          # its data operand is the placeholder the s_mov just wrote: the issue itself
                srcs = ops if op.startswith(("s_cmp", "s_bitcmp", "buffer_store", "global_store", "ds_write")) else ops[1:]
                return any(R in _sgprs(o) for o in srcs)
            # "a ticket may be in flight in R": set by the issue, cleared by an lgkmcnt(0) wait (the ticket has landed: R then holds an
            # ordinary value) or by another definition of R; propagated along fall-through and branch edges to a fixed point.  The walk
            # is path-insensitive (it also follows arm combinations the `ri == 0` tests exclude), which can only add windows, never hide one.
            def step(t, state):
                if t.startswith("s_atomic_add s%d," % R) or t == mov:
                    return True
                if (t.startswith("s_waitcnt") and "lgkmcnt(0)" in t) or writes_R(t):
                    return False
                return state
            pend_in = [False] * len(blocks)
            changed = True
            while changed:
                changed = False
                for b, insts in enumerate(blocks):
                    state = pend_in[b]
                    for t in insts:
                        state = step(t, state)
                    for nb in succ[b]:
                        if state and not pend_in[nb]:
                            pend_in[nb] = True
                            changed = True
            for b, insts in enumerate(blocks):
                state = pend_in[b]
                for k, t in enumerate(insts):
                    if t.startswith("s_atomic_add s%d," % R):
                        assert k > 0 and insts[k - 1] == mov, (who, "the placeholder is not written directly before the atomic", insts[max(0, k - 2):k + 1])
                    if reads_R(t):
                        assert not state, (who, "s%d is read where a ticket may still be in flight" % R, t, insts[max(0, k - 6):k + 1])
                        consumed = False                             # the consuming read: directly behind the wait, in its block
                        for tj in reversed(insts[:k]):
                            if tj.startswith("s_waitcnt") and "lgkmcnt(0)" in tj:
                                consumed = True
                                break
                            if writes_R(tj) or tj.startswith("s_atomic_add"):
                                break
                        ticket_reads += 1 if consumed else 0
                    state = step(t, state)
    assert ticket_reads >= 10                                         # the entry ticket's and the loop ticket's consumption, all 5 kernels


def test_fused_cloud_kernel_registers_match_the_plan(tmp_path):
    """slx_cloud_fused_kernel (csrc/slx_cloud.hip): its look-back waits for the sibling parts of a column group, so the plan
    (slx_cloud_fused_plan, csrc/slx_plan.cpp) must not count on more resident workgroups than the compiled code allows: 8-wave
    workgroups, 3 per CU with 4 chunks per part (<= 80 VGPRs: 6 waves per SIMD), 2 per CU with 7 (<= 96... 102: 5 waves per SIMD);
    no spills, no scratch."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is not installed")
    out = str(tmp_path / "slx_cloud.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, "slx_cloud.hip"), "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    seen = 0
    for blk in re.split(r"\n  - \.agpr_count:", meta):
        m = re.search(r"\.name:\s+\S*slx_cloud_fused_kernelILb(\d)ELb(\d)ELj(\d)E", blk)
        if not m:
            continue
        seen += 1
        v = {f: int(re.search(r"\.%s:\s+(\d+)" % f, blk).group(1)) for f in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size")}
        assert v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (m.groups(), v)
        waves_per_simd = 512 // ((v["vgpr_count"] + 7) // 8 * 8)
        need = 6 if m.group(3) == "4" else 4                          # 3 / 2 workgroups of 8 waves per CU = 24 / 16 waves on 4 SIMDs
        assert waves_per_simd >= need, (m.groups(), v["vgpr_count"], waves_per_simd)
    assert seen == 8


def test_tracker_kernel_keeps_a_whole_frame_resident(tmp_path):
    """slx_track_fused_kernel<10> (csrc/slx_track.hip): the 1 200 workgroups of a 1920 x 1200 frame run as ONE wave of dependent chains (DESIGN.md
    section 7) -- 4.7 per CU, so the compiled code must allow 5 four-wave workgroups per CU: <= 96 VGPRs (5 waves per SIMD), <= 32 KiB of LDS, and
    neither spills nor scratch; the sliding sums come in by 16-bit loads, two columns per lane (30 per lane, no byte loads of the image)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is not installed")
    out = str(tmp_path / "slx_track.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, "slx_track.hip"), "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    blk = [b for b in re.split(r"\n  - \.agpr_count:", meta) if re.search(r"\.name:\s+\S*slx_track_fused_kernelILi10E", b)]
    assert len(blk) == 1
    v = {f: int(re.search(r"\.%s:\s+(\d+)" % f, blk[0]).group(1)) for f in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    assert v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, v
    assert 512 // ((v["vgpr_count"] + 7) // 8 * 8) >= 5 and v["group_segment_fixed_size"] <= 32 * 1024, v
    name = re.search(r"\.name:\s+(\S*slx_track_fused_kernelILi10E\S*)", blk[0]).group(1)
    body = text[text.index("\n" + name + ":"):]
    body = body[:body.index("s_endpgm")]
    assert len(re.findall(r"global_load_ushort", body)) == 30 and not re.findall(r"global_load_ubyte", body)
