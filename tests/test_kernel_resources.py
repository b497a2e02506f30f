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
        if GB == 9:
            continue                                   # the stream kernel: planned for 4 waves per SIMD, checked by the <= 128 test above
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
    (csrc/slx_kernels.hip: fetch_issue / fetch_wait).  Neither hipcc's register allocator nor its waitcnt insertion knows the register
    is pending in between: a copy, spill or reuse of it there would read the placeholder (1) instead of the ticket, and rows would be
    skipped or decoded twice with no error.  This pins the compiled code of every instantiation:
      * all s_atomic_add of the kernel write ONE register R, each directly behind `s_mov_b32 R, 1`;
      * nothing else ever writes R;
      * every read of R sits in a basic block where an `s_waitcnt ... lgkmcnt(0)` comes between the block's start (or the block's
        own s_atomic_add) and the read -- so no instruction can see R between an issue and its wait, on any path."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is not installed")
    out = str(tmp_path / "slx_kernels.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, "slx_kernels.hip"), "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    checked = 0
    for F in (1, 2, 3, 4):
        start = next(i for i, ln in enumerate(lines) if re.match(r"^_ZN\S*slx_stream_kernelILi%dEEEv10SlxKParams:" % F, ln))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = [ln.split(";")[0].rstrip() for ln in lines[start + 1:end]]
        insts = [(i, ln.strip()) for i, ln in enumerate(body) if ln.startswith("\t") and not ln.strip().startswith(".")]
        labels = {i for i, ln in enumerate(body) if re.match(r"^\.LBB\S*:", ln)}
        atomics = [(i, t) for i, t in insts if t.startswith("s_atomic_add ")]
        assert len(atomics) >= 2, (F, "expected the entry ticket and the loop's ticket")
        dest = {re.match(r"s_atomic_add s(\d+),", t).group(1) for _, t in atomics}
        assert len(dest) == 1, (F, "tickets land in different registers: a phi / copy of the pending register", dest)
        R = int(dest.pop())
        pos = {i: k for k, (i, _) in enumerate(insts)}
        for i, t in atomics:
            prev = insts[pos[i] - 1][1]
            assert prev == "s_mov_b32 s%d, 1" % R, (F, prev, t)
        for k, (i, t) in enumerate(insts):
            op, _, rest = t.partition(" ")
            operands = [o.strip() for o in rest.split(",")] if rest else []
            if not any(R in _sgprs(o) for o in operands):
                continue
            if t == "s_mov_b32 s%d, 1" % R or t.startswith("s_atomic_add s%d," % R):
                continue
            # scalar / vector ALU: operand 0 is the destination (compares to vcc / scc name it explicitly as well)
            assert R not in _sgprs(operands[0]) or op.startswith(("s_cmp", "v_cmp", "s_bitcmp")), (F, "something else writes the ticket register", t)
            # a read: walk back inside the basic block; an lgkmcnt(0) wait must come before the block's start or an s_atomic_add does
            waited = False
            for kk in range(k - 1, -1, -1):
                j, tj = insts[kk]
                if any(lbl > j and lbl <= i for lbl in labels):
                    break                                             # left the basic block
                if tj.startswith("s_atomic_add"):
                    break
                if tj.startswith("s_waitcnt") and "lgkmcnt(0)" in tj:
                    waited = True
                    break
                if tj.startswith(("s_cbranch", "s_branch")):
                    break
            assert waited, (F, "the ticket register is read with no lgkmcnt(0) wait before it in its block", t, i)
            checked += 1
    assert checked >= 8                                               # two reads (entry, loop) per instantiation at least
