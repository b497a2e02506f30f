// slx_plan.cpp -- host-side planning of a decode launch: which kernel, how the frame-sets are cut into work items, the grid.
//
// Nothing here touches the GPU or the HIP runtime: the same translation unit is compiled into libslx.so and, by g++ with
// -fsanitize=address,undefined, into tests/cpp/host_sanitize (sanitizers on the CPU side), which walks
// many tile shapes through slx_plan_launch and checks that the items of every plan cover every row of every frame-set
// exactly once and that no 32-bit offset the kernel forms can wrap.  slx_kernels.hip holds the kernels and launches a plan.
#include <algorithm>
#include <cstdint>
#include <initializer_list>

#include "slx_kernels.h"

namespace {
constexpr int kInvalid = 1;            // hipErrorInvalidValue: what slx_launch_fused hands back for a plan that cannot be made
constexpr int kCloudTile = 64;         // slx_cloud_* kernels: tile edge (slx_kernels.hip)
}  // namespace

int slx_num_variants(void) { return 4; }

int slx_cloud_entries(int width, int height) { return width * ((height + kCloudTile - 1) / kCloudTile); }

int slx_cloud_tiles(int width, int height) { return ((width + kCloudTile - 1) / kCloudTile) * ((height + kCloudTile - 1) / kCloudTile); }

// ---- fused point cloud (slx_cloud.hip) ----
// A part is 16 columns x R rows; R = 256 (the workgroup's 512 lanes x 4 loads of 16 bytes, all in flight), less for a
// shorter map (rounded up to the 32 rows a pass of the lanes covers), more only when 16 parts of 256 rows do not reach the bottom.
size_t slx_cloud_fused_lds_bytes(int rows_per_part) { return (size_t)rows_per_part * 16u * sizeof(double) + (SLX_CLOUD_THREADS / 64u) * 192u * sizeof(double); }

size_t slx_cloud_fused_words(int groups, int parts) { return (size_t)SLX_CLOUD_COUNTERS * 16u + (size_t)groups * (size_t)parts * 17u; }

bool slx_cloud_fused_plan(int W, int H, unsigned n_cus, int *groups, int *parts, int *rows_per_part)
{
    if (W < 1 || H < 1 || (unsigned long long)W * (unsigned)H >= (1ull << 31)) return false;   // 32-bit point counts and offsets
#ifndef SLX_CLOUD_PART_ROWS
#define SLX_CLOUD_PART_ROWS 256      /* experiments: -DSLX_CLOUD_PART_ROWS=128 | 192 (tools/cloud_bench.py --lib) */
#endif
    int R = H >= SLX_CLOUD_PART_ROWS ? SLX_CLOUD_PART_ROWS : (H + 63) / 64 * 64;
    if ((H + R - 1) / R > 16) R = ((H + 15) / 16 + 63) / 64 * 64;
    const int P = (H + R - 1) / R;
    const size_t lds = slx_cloud_fused_lds_bytes(R) + 2048u;        // + the kernel's static words
    if (P > 16 || R > SLX_CLOUD_MAX_ROWS || lds > 64u * 1024u) return false;   // (64 KiB: the dynamic LDS a launch gets without asking for more)
    // the look-back waits for the sibling parts of a column group, whose tickets are adjacent: they must be able to be resident together
    const unsigned long long cus = n_cus ? n_cus : 256u;
    // workgroups of 8 waves a CU keeps: by LDS, and by registers (74 VGPRs with 4 chunks per part: 6 waves per SIMD = 3 workgroups; 98 with
    // 7 chunks: 5 waves per SIMD = 2 workgroups)
    const unsigned long long by_regs = R > 256 ? 2u : 3u;
    // ... PER XCD: the kernel numbers its workgroups in start order within classes of indices, and what the dispatcher keeps in index order is
    // each XCD's share of them (an eighth of the compute units; a device of fewer than 8 counts as one) -- the sibling parts of a column
    // group must fit into ONE XCD's resident set whichever XCD they land on.  (Should the assumption fail all the same, the look-back's
    // polls are bounded and the host repeats the frame on the two-launch path: slx_cloud.hip.)
    const unsigned long long per_xcd = std::max<unsigned long long>(1ull, cus / 8ull) * std::min<unsigned long long>(by_regs, 160u * 1024u / lds);
    if (per_xcd < (unsigned long long)P) return false;
    const long long G = ((long long)W + 15) / 16;
    if (G * P >= (1ll << 24)) return false;
    *groups = (int)G;
    *parts = P;
    *rows_per_part = R;
    return true;
}

bool slx_fast_arith_ok(const SlxKParams &kp)
{
    for (int f = 0; f < kp.n_freq; f++)
        if (kp.period[f] > (1 << 14)) return false;
    // the depth quotient's operands must stay far inside the double range (tri_depth<LEAN>)
    const double big = 0x1p90;
    for (double v : {kp.cA, kp.cB, kp.K1, kp.K2, kp.P00, kp.P01, kp.P20, kp.P21, kp.fu, kp.fv, kp.cx, kp.cy})
        if (!(__builtin_fabs(v) < big)) return false;
    return true;
}

// The planes one ring chunk of the strip kernels holds must be equally spaced, ascending, the last within 2 GiB of the first
// (32-bit buffer offsets): the layout of a batch and of the context's staging slab; anything else takes the generic kernel.
static bool planes_equally_spaced(const uint8_t *const *pl, int np)
{
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(pl[0]);
    const uintptr_t step = np > 1 ? reinterpret_cast<uintptr_t>(pl[1]) - a0 : 0;
    if (np > 1 && reinterpret_cast<uintptr_t>(pl[1]) < a0) return false;
    for (int k = 0; k < np; k++)
        if (reinterpret_cast<uintptr_t>(pl[k]) != a0 + (uintptr_t)k * step) return false;
    return (unsigned long long)step * (unsigned)(np > 1 ? np - 1 : 0) < (1ull << 31);
}

// The work-item geometry's 32-bit arithmetic (quads per row, plane offsets, output byte offsets with the rows past the tile).
static bool strip_geometry_ok(const SlxKParams &kp)
{
    if (kp.quads_per_row == 0 || kp.quads_per_row > 1024) return false;
    if ((unsigned long long)kp.row_stride * (unsigned)kp.height >= (1ull << 31)) return false;
    return (unsigned long long)kp.width * ((unsigned)kp.height + 2048ull) < (1ull << 29);
}

bool slx_strip_eligible(const SlxKParams &kp, int mode, bool aux)
{
    if (!kp.aligned) return false;
    if (mode == SLX_MODE_PHASE_ONLY) {
        // slx_decoder_strip_kernel: the reference's decoder, 4 steps (R/CDecodePhase.cpp:59-62), a period the f32 identities were swept for
        return !aux && kp.pix && kp.n_freq == 1 && kp.n_steps == 4 && kp.period[0] <= (1 << 14) && strip_geometry_ok(kp) && planes_equally_spaced(kp.phase, 4);
    }
    if (mode == SLX_MODE_GRAY_ONLY) {
        // 6 bits (the reference's count, R/StaticParameters.cpp:16) with the reflected-code table (R/Patterns/vGrayCode.txt)
        return !aux && kp.gray_out && kp.gray_bits == 6 && kp.std_gray && strip_geometry_ok(kp) && planes_equally_spaced(kp.gray, 12);
    }
    if (aux && (kp.pix || kp.gray_out)) return false;                // the per-frequency pix planes and the Gray plane come from the generic kernel only
    {
        // second-pass divisors (x = z uc / fu, y = z vc / fv) must sit well inside the double range for the unscaled division
        const double lo = 0x1p-90;
        if (aux && !(__builtin_fabs(kp.fu) > lo && __builtin_fabs(kp.fv) > lo)) return false;
    }
    if (kp.n_steps != 4) {                                           // x1 fast path: 8 steps, Gray-free, the expected weight table
        if (!(kp.n_steps == 8 && mode == SLX_MODE_MULTIFREQ)) return false;
        const float r = kp.wy[1];
        const float ey[8] = {1.f, r, 0.f, -r, -1.f, -r, 0.f, r}, ex[8] = {0.f, r, 1.f, r, 0.f, -r, -1.f, -r};
        for (int k = 0; k < 8; k++)
            if (kp.wy[k] != ey[k] || kp.wx[k] != ex[k]) return false;
        if (!(r > 0.70f && r < 0.71f) || kp.wscale != 0.25f) return false;
    }
    if (mode != SLX_MODE_MULTIFREQ && mode != SLX_MODE_GRAY_PHASE && mode != SLX_MODE_MULTIFREQ_GRAYMASK) return false;
    if (!slx_fast_arith_ok(kp)) return false;
    return strip_geometry_ok(kp) && planes_equally_spaced(kp.phase, kp.n_freq * kp.n_steps);
}

// Waves per SIMD the register file allows a strip-kernel instantiation (512 VGPRs per lane and SIMD, allocated in blocks of 8):
// 4 up to 128 VGPRs, 3 up to 168.  The instantiations above 128 are the optional-plane (AUX) variants of the wide configurations;
// tests/test_kernel_resources.py compiles the kernels and fails when this table claims more waves than the compiled code allows,
// when any instantiation spills a vector register or uses scratch (the counted s_waitcnt vmcnt(n) of the DMA ring are exact only
// while hipcc adds no vector-memory operation of its own), or when a Gray-free / ring-Gray instantiation without the optional
// planes leaves the 128 that 4 waves per SIMD need.
unsigned slx_strip_waves_per_simd(int mode, int n_freq, int gray_ring_bits, int n_steps, int aux)
{
    if (mode == SLX_MODE_PHASE_ONLY || mode == SLX_MODE_GRAY_ONLY) return 8;   // slx_decoder_strip_kernel: <= 64 VGPRs
    if (!aux) return 4;
    if (mode == SLX_MODE_MULTIFREQ) return n_freq >= 4 ? 3 : 4;
    if (mode == SLX_MODE_MULTIFREQ_GRAYMASK) return (gray_ring_bits ? n_freq >= 2 : n_freq >= 3) ? 3 : 4;
    (void)n_steps;
    return 4;
}

// Rows per work item of the strip kernel, from the measured sweeps of tools/single_set.py (profiles/r03_rows_sweep.json):
//  * a launch that fills the chip several times over (>= 15 360 items = 3.75 per resident wave slot) takes the kernel's
//    preferred item: 16 rows for the Gray-free 4-step kernels (VALU-bound; 16 rows amortise an item's start-up best and
//    the tiers below cut the drain), 3 rows where the Gray planes ride the ring (REF 166 vs 176 us at 8 rows and 220 at 16,
//    C3 218 vs 226 / 238: those kernels wait for their DMA, and short items keep the waves of a SIMD out of step), 10 rows
//    for 8 steps (C5: 368 vs 407 us at 16);
//  * a launch of the VALU-bound kernels too small for that gets ONE round of items when items of <= 10 rows can cover it -- the smallest item count
//    that fits the resident wave slots, so every slot works from the first cycle to the last and nothing is left for a
//    thinly filled second round: one 1920x1200 frame-set (9 000 wave-rows for 4 096 slots; the call the reference makes,
//    R/CCalculation.cpp:171-206) runs as 3 000 items of 3 rows in 13.2 us, against 14.7 us as 9 000 items of one row;
//  * in between, the largest item that still gives 15 360 items.
unsigned slx_strip_rows_model(unsigned height, unsigned interleave, unsigned chunks_per_group, unsigned n_sets, unsigned slots_per_cu, unsigned preferred,
                              unsigned n_cus)
{
    const unsigned long long cus = n_cus ? n_cus : 256u;             // the device's compute units (a partitioned or smaller device has fewer)
    const unsigned long long slots = cus * std::max(1u, slots_per_cu), many = cus * 20ull * 3ull;
    auto items = [&](unsigned r) {
        const unsigned long long rows_group = (unsigned long long)interleave * r;
        return ((height + rows_group - 1) / rows_group) * chunks_per_group * (unsigned long long)n_sets;
    };
    const unsigned cand[] = {16, 12, 10, 8, 6, 5, 4, 3, 2, 1};
    if (preferred < 1 || preferred > 16) preferred = 16;
    if (items(preferred) >= many) return preferred;
    if (preferred <= 3) {
        // the DMA-bound Gray kernels like short items whatever the launch: 3 rows while that gives 1.5 items per slot
        // (4 frame-sets of the reference's size: 27.1 us, against 29.6 us as one round of 6-row items), else 2, else 1
        // (one frame-set at 1280x1024: 2 560 items of 2 rows; C3's single frame-set 20.6 us at 2 rows, 21.7 at 3)
        if (items(3) * 2 >= slots * 3) return 3;
        return items(2) * 2 >= slots ? 2 : 1;
    }
    unsigned one_round = 0;                                          // smallest item that covers the launch in one round
    for (unsigned r : cand)
        if (items(r) <= slots) one_round = r;
    if (one_round >= 1 && one_round <= 10) return one_round;
    for (unsigned r : cand)
        if (r <= preferred && items(r) >= many) return r;
    return 1;
}

int slx_plan_launch(const SlxKParams &kp_in, int mode, bool aux, int n_sets, int variant, const SlxTuning *tune, SlxLaunchPlan *plan)
{
    if (!plan) return kInvalid;
    plan->mode = mode;
    plan->aux = aux ? 1 : 0;
    plan->gray_ring_bits = 0;
    plan->stream = 0;
    const SlxTuning none{};
    const SlxTuning &tn = tune ? *tune : none;
    const bool can_strip = slx_strip_eligible(kp_in, mode, aux);
    if (variant == SLX_VARIANT_GENERIC || variant == SLX_VARIANT_GENERIC_FAST || !can_strip) {
        if (variant == SLX_VARIANT_STRIP) return kInvalid;
        SlxKParams &kg = plan->kp;
        kg = kp_in;
        kg.fast_arith = (variant != SLX_VARIANT_GENERIC && mode >= SLX_MODE_GRAY_PHASE && slx_fast_arith_ok(kp_in)) ? 1 : 0;
        if (variant == SLX_VARIANT_GENERIC_FAST && !kg.fast_arith) return kInvalid;
        // generic kernel: one lane per quad (the Gray-mask mode: 62 quads + 2 halo lanes per wave), one grid row per frame-set
        unsigned long long threads = kg.n_quads;
        if (mode == SLX_MODE_MULTIFREQ_GRAYMASK) threads = (((unsigned long long)kg.n_quads + 61ull) / 62ull) * 64ull;
        const unsigned long long gx = (threads + 255ull) / 256ull;
        if (gx == 0 || gx >= (1ull << 31) || n_sets <= 0 || n_sets > 65535) return kInvalid;
        plan->strip = 0;
        plan->grid_x = (unsigned)gx;
        plan->grid_y = (unsigned)n_sets;
        plan->block = 256;
        plan->lds_bytes = 0;
        return 0;
    }
    SlxKParams &kp = plan->kp;
    kp = kp_in;
    const bool decoder = mode == SLX_MODE_PHASE_ONLY || mode == SLX_MODE_GRAY_ONLY;      // slx_decoder_strip_kernel
    if (mode == SLX_MODE_GRAY_ONLY) {
        // the decoder kernel finds its planes behind the phase-plane fields whichever decoder it is
        kp.plane_base = kp.gray[0];
        kp.phase_step = (unsigned)(kp.gray[1] - kp.gray[0]);
        kp.phase_set_stride = kp.gray_set_stride;
    } else {
        kp.plane_base = kp.phase[0];                                 // equally spaced, ascending (slx_strip_eligible)
        kp.phase_step = kp.n_freq * kp.n_steps > 1 ? (unsigned)(kp.phase[1] - kp.phase[0]) : 0u;
    }
    kp.phase_first = 0;
    kp.gray_step = 0;
    kp.dma_imm = ((!decoder && kp.n_freq * kp.n_steps == 1) || kp.phase_step >= 256u) ? 1 : 0;
    // geometry: `interleave` rows end to end fill whole waves; an item is 64 quads x rows_per_lane rows
    const unsigned QR = kp.quads_per_row;
    unsigned g = QR, h = 64;
    while (h) { const unsigned r = g % h; g = h; h = r; }          // gcd(QR, 64)
    kp.interleave = 64u / g;
    // Weave: more rows per row group than whole waves need -- a lane's rows are then `interleave` apart and the waves of a group
    // sweep that many consecutive rows in step.  Measured (tools/ab.py, round 4, two boxes): 8 rows instead of 2 on the 1920-wide
    // 4-step kernels, config 4 282.9 -> 277.1 us (-2.0 %; the other box 290.1 -> 285.8), config 3 215.5 -> 209.9 (-2.6 %); 16 rows
    // lose it again; the 1280-wide cases (one row = 5 whole chunks) lose 0.5-1.7 % with any weave and are left alone, 4096-wide
    // 8-step within noise.  Only full frames: a row group of 8 x rows-per-item rows wastes too much of a short tile.
    unsigned weave = 0;
    if (tn.weave > 1) weave = (unsigned)tn.weave;
    else if (tn.weave == 0 && !decoder && kp.interleave == 2 && kp.n_steps == 4 && kp.height >= 512) weave = 8;
    if (weave > 1) {
        const unsigned m = std::min(64u, weave) / kp.interleave;
        if (m > 1) kp.interleave *= m;
    }
    kp.chunks_per_group = mode == SLX_MODE_MULTIFREQ_GRAYMASK ? (kp.interleave * QR + 61u) / 62u   // 62 quads + 2 halo lanes per wave
                                                             : kp.interleave * QR / 64u;
    kp.plain_order = tn.plain_order ? 1 : 0;
    // Gray planes ride the DMA ring when there are 6 bits of them (the reference's and config 3's count) and they are equally
    // spaced, ascending, the last within 2 GiB of the first (the kernel addresses them through a descriptor of their own, based at
    // Gray plane 0 of the frame-set: where the phase planes live does not matter); otherwise the kernel reads them with ordinary loads
    int gb = 0;
    if (!decoder && mode != SLX_MODE_MULTIFREQ && kp.gray_bits == 6 && !tn.gray_plain && planes_equally_spaced(kp.gray, 12)) {
        gb = 6;
        kp.gray_step = (unsigned)(kp.gray[1] - kp.gray[0]);
        if (kp.gray_step < 256u) kp.dma_imm = 0;
    }
    // Stream kernel (slx_kernels.hip: slx_stream_kernel): the Gray-free 4-step depth-only class, launches that fill the chip many times over.
    // (An instantiation with the optional planes x, y, U, k was built and measured in round 5 -- C4 x 8 / 16 / 32 frame-sets: +1.5 / -1.6 /
    // -0.6 % against the strip kernel on the same box -- and is not in the tree: profiles/experiments/r05_stream_kernel_optional_planes.patch.)
    plan->stream = 0;
    if (!decoder && mode == SLX_MODE_MULTIFREQ && kp.n_steps == 4 && !aux && kp.sq_counters && tn.stream != 1 && tn.weave <= 1) {
        const unsigned il = 64u / g;                                     // no weave: the queues hand the rows out in order anyway
        const unsigned cpg = il * QR / 64u;
        const unsigned R = (tn.stream_rows >= 2 && tn.stream_rows <= 16) ? (unsigned)tn.stream_rows : 2u;   // measured: 2 rows 270, 4: 274, 8: 284, 16: 302 us (C4 x 32)
        const unsigned gps = ((unsigned)kp.height + R * il - 1u) / (R * il);
        const unsigned long long groups_total = (unsigned long long)gps * (unsigned)n_sets;
        const unsigned cus = kp.n_cus ? kp.n_cus : 256u;
        // resident waves per CU: 16 (4 per SIMD) in 4-wave workgroups; experiments: strip_waves = w makes w-wave workgroups and as many
        // of them as the CU's 160 KiB of LDS hold (1-wave workgroups: 20 waves per CU = 5 per SIMD)
        const unsigned lds_w0 = 2u * (unsigned)kp.n_freq * 4u * 256u + 2048u;
        const unsigned wpw = (tn.strip_waves >= 1 && tn.strip_waves <= 4) ? (unsigned)tn.strip_waves : 4u;
        const unsigned per_cu = std::min(32u, wpw * (160u * 1024u / (wpw * lds_w0)));
        const unsigned long long waves = (unsigned long long)cus * (tn.strip_waves ? per_cu : 16u);
        // Queues: a wave polls queue (its number) % queues, so every queue that holds items needs a wave of its own residue -- at most
        // as many queues as the launch has waves (a partitioned or small device: fewer than 255 resident waves), else a queue's row
        // groups would never be decoded
        const unsigned m = cpg <= 255u ? (unsigned)std::min<unsigned long long>(255u / cpg, waves / cpg) : 0u;
        // >= 5 items per resident wave.  Same-box sweep of round 5 (tools/stream_sweep.sh, C4 x 4 ... 31 frame-sets, stream against strip
        // kernel): 4 sets -3 % (the minima equal), 6 sets -6 %, 8 -10 %, 10 -14 %, 12 -10 %, 16 -7 %, 20 ... 24 -6 %, 28 ... 31 -3 %: the
        // queues pay from about 5 items per wave on (round 4's threshold was 8)
        const bool big = groups_total * cpg >= 5ull * waves;
        if (m >= 1 && cpg * m <= SLX_STREAM_MAX_QUEUES && groups_total * gps < (1ull << 32) && groups_total < (1ull << 31) &&
            (tn.stream == 2 ? groups_total * cpg >= 1 : big)) {
            kp.interleave = il;
            kp.chunks_per_group = cpg;
            kp.sq_queues = cpg * m;
            kp.sq_m = m;
            kp.sq_rows = R;
            kp.sq_groups_per_set = gps;
            kp.sq_groups_total = (unsigned)groups_total;
            kp.sq_magic = (unsigned)((1ull << 32) / gps) + 1u;
            kp.n_tiers = 0;
            const unsigned lds_w = lds_w0;
            const unsigned long long items = groups_total * cpg;
            const unsigned long long want_waves = std::min<unsigned long long>(waves, items);
            plan->strip = 1;
            plan->stream = 1;
            plan->gray_ring_bits = 0;
            plan->block = 64u * wpw;
            plan->grid_x = (unsigned)((want_waves + wpw - 1ull) / wpw);
            plan->grid_y = 1;
            plan->lds_bytes = wpw * lds_w;
            return 0;
        }
    }
    // Stream kernel of the reference's own mode (slx_gstream_kernel, round 6): 6 Gray bits on the ring + one 4-step frequency, depth only.
    // Same queues and counters as slx_stream_kernel; a row is two ring chunks there, an item ONE row (REF x 32: 164.3 us with 1-row
    // items, 171.3 with 2, 177.6 with 3: profiles/r06_gstream_ab.log).  Taken from 8 items per resident wave on -- same-box against the
    // strip kernel: 32 frame-sets of the reference's size -3.1 %, 16 -3.5 %, 8 -2.4 %, 4 (5 items per wave) +0.7 %.
    if (!decoder && gb == 6 && mode == SLX_MODE_GRAY_PHASE && kp.n_steps == 4 && !aux && kp.sq_counters && tn.stream != 1 && tn.weave <= 1) {
        const unsigned il = 64u / g, cpg = il * QR / 64u;
        const unsigned R = (tn.stream_rows >= 1 && tn.stream_rows <= 16) ? (unsigned)tn.stream_rows : 1u;
        const unsigned gps = ((unsigned)kp.height + R * il - 1u) / (R * il);
        const unsigned long long groups_total = (unsigned long long)gps * (unsigned)n_sets;
        const unsigned cus = kp.n_cus ? kp.n_cus : 256u;
        const unsigned lds_w = 2u * 12u * 256u + 2048u;
        const unsigned wpw = (tn.strip_waves >= 1 && tn.strip_waves <= 4) ? (unsigned)tn.strip_waves : 4u;
        const unsigned long long waves = (unsigned long long)cus * 16u;                     // 4 per SIMD: <= 128 VGPRs (tests/test_kernel_resources.py)
        const unsigned m = (cpg >= 1 && cpg <= 255u) ? (unsigned)std::min<unsigned long long>(255u / cpg, waves / cpg) : 0u;
        const bool big = groups_total * cpg >= 8ull * waves;
        if (m >= 1 && cpg * m <= SLX_STREAM_MAX_QUEUES && groups_total * gps < (1ull << 32) && groups_total < (1ull << 31) &&
            (tn.stream == 2 ? groups_total * cpg >= 1 : big)) {
            kp.interleave = il;
            kp.chunks_per_group = cpg;
            kp.sq_queues = cpg * m;
            kp.sq_m = m;
            kp.sq_rows = R;
            kp.sq_groups_per_set = gps;
            kp.sq_groups_total = (unsigned)groups_total;
            kp.sq_magic = (unsigned)((1ull << 32) / gps) + 1u;
            kp.n_tiers = 0;
            const unsigned long long items = groups_total * cpg;
            const unsigned long long want_waves = std::min<unsigned long long>(waves, items);
            plan->strip = 1;
            plan->stream = 2;
            plan->gray_ring_bits = gb;
            plan->block = 64u * wpw;
            plan->grid_x = (unsigned)((want_waves + wpw - 1ull) / wpw);
            plan->grid_y = 1;
            plan->lds_bytes = wpw * lds_w;
            return 0;
        }
    }
    // LDS per wave: 2 ring slots (max(4 n_freq, 2 gb) planes with 4 steps, 8 planes with 8 steps, 256 B each) + 2 KiB of depth staging
    const unsigned ring_planes = mode == SLX_MODE_GRAY_ONLY ? 12u : kp.n_steps == 4 ? std::max((unsigned)kp.n_freq * 4u, 2u * (unsigned)gb) : 8u;
    const unsigned lds_wave = 2u * ring_planes * 256u + 2048u + (aux ? 2048u : 0u);      // + the optional planes' staging area
    // rows per item (slx_strip_rows_model): the kernel's preferred item for a launch that fills the chip many times over,
    // one round of items for a small one
    // resident waves per CU: what the LDS holds, and what the instantiation's registers allow (slx_strip_waves_per_simd)
    const unsigned waves_by_regs = 4u * slx_strip_waves_per_simd(mode, kp.n_freq, gb, kp.n_steps, aux ? 1 : 0);
    const unsigned slots_per_cu = std::min(waves_by_regs, 160u * 1024u / lds_wave);
    const unsigned preferred = decoder ? 3u : kp.n_steps == 8 ? 10u : gb ? 3u : 16u;   // the decoder kernels move bytes and little else: short items
    unsigned rb = slx_strip_rows_model((unsigned)kp.height, kp.interleave, kp.chunks_per_group, (unsigned)n_sets, slots_per_cu, preferred, kp.n_cus);
    // <= 32 rows: slx_strip_eligible bounds the 32-bit output offsets for interleave (<= 64) x 32 rows past the tile
    if (tn.strip_rows >= 1 && tn.strip_rows <= 32) rb = (unsigned)tn.strip_rows;
    // Tiers: the head of every frame-set in items of rb rows, then shorter items (a quarter of the previous tier's rows)
    // for the last tail_pct % of the rows; with more than two tiers each takes 60 % of what is left, the last one all of
    // it.  Short items run last and cut the end of the launch, where the chip drains for about one item's lifetime.
    unsigned tail_pct = 20, tail_rb = rb / 4, tiers = 2;   // measured (tools/ab.py, C4 and C3): 2 tiers beat 1 by 2 %, 3 and 4 lose it again
    if (tn.tail_pct != 0) tail_pct = tn.tail_pct < 0 ? 0u : (unsigned)tn.tail_pct;
    if (tn.tail_rows > 0) tail_rb = (unsigned)tn.tail_rows;
    if (tn.tiers >= 1 && tn.tiers <= SLX_MAX_TIERS) tiers = (unsigned)tn.tiers;
    const unsigned rows_group = kp.interleave * rb;
    const unsigned groups = ((unsigned)kp.height + rows_group - 1) / rows_group;
    if (!(rb >= 8 && tail_rb >= 1 && tail_rb < rb && tail_pct > 0 && tail_pct < 100 && groups >= 4)) tiers = 1;   // (tiers below 8 rows: tried on REF and C3 at 3 rows, no gain)
    // waves per workgroup: as many as keep the most waves resident in the CU's 160 KiB of LDS (16 at most: 4 per SIMD)
    const unsigned lds_shared = 0u;                                     // nothing is shared between the waves of a workgroup
    unsigned waves_per_wg = 4u;
    {
        unsigned best = 0;
        for (unsigned w = 4; w >= 1; w--) {
            const unsigned resident = std::min(waves_by_regs, w * (160u * 1024u / (w * lds_wave + lds_shared)));
            if (resident > best) { best = resident; waves_per_wg = w; }
        }
    }
    if (tn.strip_waves >= 1 && tn.strip_waves <= 4) waves_per_wg = (unsigned)tn.strip_waves;
    unsigned long long need_wgs = 0;
    {
        unsigned row0 = 0, t = 0, r = rb;
        unsigned head_groups = tiers > 1 ? (unsigned)((unsigned long long)groups * (100u - tail_pct) / 100u) : groups;
        if (head_groups < 1 || head_groups >= groups) { head_groups = groups; tiers = 1; }
        unsigned long long first_wg = 0;
        while (true) {
            const unsigned group_rows = kp.interleave * r;
            const unsigned left = (unsigned)kp.height - row0;
            unsigned g_here;
            if (t == 0) g_here = head_groups;
            else if (t + 1 == tiers || r == 1) g_here = (left + group_rows - 1) / group_rows;        // the last tier takes what is left
            else g_here = std::max(1u, (unsigned)((unsigned long long)left * 60u / 100u / group_rows));
            if ((unsigned long long)g_here * group_rows >= left) g_here = (left + group_rows - 1) / group_rows;
            kp.tier_rows[t] = r;
            kp.tier_row0[t] = row0;
            kp.tier_items_per_set[t] = g_here * kp.chunks_per_group;
            const unsigned long long items = (unsigned long long)kp.tier_items_per_set[t] * (unsigned)n_sets;
            const unsigned long long wgs = (items + waves_per_wg - 1) / waves_per_wg;
            if (items >= (1ull << 32) || first_wg + wgs >= (1ull << 31)) return kInvalid;
            kp.tier_items[t] = (unsigned)items;
            kp.tier_first_wg[t] = (unsigned)first_wg;
            kp.tier_wgs[t] = (unsigned)wgs;
            first_wg += wgs;
            row0 += g_here * group_rows;
            t++;
            if (row0 >= (unsigned)kp.height || t == SLX_MAX_TIERS) break;
            r = t == 1 ? tail_rb : std::max(1u, r / 4u);
        }
        kp.n_tiers = t;
        need_wgs = first_wg;                                             // the waves past a tier's last item idle
    }
    const unsigned threads = waves_per_wg * 64u;
    if (need_wgs == 0 || need_wgs >= (1ull << 31)) return kInvalid;
    size_t lds = (size_t)waves_per_wg * lds_wave + lds_shared;
    if (tn.lds_pad_kib > 0 && tn.lds_pad_kib <= 128) lds += (size_t)tn.lds_pad_kib * 1024u;   // experiments: lower the occupancy
    if (lds > 160u * 1024u) return kInvalid;
    plan->strip = 1;
    plan->gray_ring_bits = gb;
    plan->grid_x = (unsigned)need_wgs;
    plan->grid_y = 1;
    plan->block = threads;
    plan->lds_bytes = lds;
    return 0;
}
