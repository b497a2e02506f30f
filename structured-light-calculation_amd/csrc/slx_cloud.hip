// slx_cloud.hip -- the point cloud of a depth map in ONE launch that reads the depth ONCE.
//
// CCalculation::Result (R/CCalculation.cpp:323-357) walks u outer / v inner and writes "x y z" for every depth inside the FOV,
// x = z (u - cx) / fu, y = z (v - cy) / fv (R/CCalculation.cpp:756-771).  The cloud is a stream compaction in COLUMN-major order of
// a ROW-major map.  Rounds 1-4 did it in two launches (slx_cloud_count_kernel, slx_cloud_write_kernel in slx_kernels.hip: the depth
// map read twice, 22.7 us for 1920 x 1200); they stay as the path for devices and shapes this kernel's plan refuses
// (slx_cloud_fused_plan, slx_plan.cpp) and as the second opinion of the tests.
//
// Here a workgroup owns a STRIP PART: 16 columns x R rows (part p of column group g; P parts per group).
//   1. the part's depths come in by rows -- 128 contiguous bytes per row, 16 bytes per lane, every load of the part in flight at
//      once -- into an LDS tile, XOR-swizzled so that the column walks of step 3 spread over the banks; on the way every lane
//      counts the kept depths of its two columns;
//   2. the part publishes its 16 column counts and their sum (epoch-tagged 64-bit words, relaxed atomic stores) and collects what it needs
//      to know where its points go: the sums of every part of the column groups before it, and the column counts of its sibling
//      parts -- a decoupled look-back over a few hundred words, no scan kernel, no second launch;
//   3. a wave takes a column of the tile, 64 rows at a time: lane = row; x and y are computed into registers WHILE the look-back of
//      step 2 waits for the other parts (3a), then the kept lanes' rank (ballot + popcount below the lane) places the point in the
//      column's run, the run is packed in LDS and leaves as contiguous doubles, 512 bytes per store (3b).
// Order of the parts: tickets number the workgroups in the order they START (64 counters, one per class of workgroup indices: see the
// kernel), ticket -> (group, part) group-major.  A workgroup then only ever waits for workgroups with a lower ticket -- which have
// started, and publish before they wait for anything -- or for its own group's parts, whose tickets are adjacent: with at least P
// resident workgroups (the plan checks the device for it) no wait can last forever, whatever else the device is running.
// Nothing is zeroed between launches: every counter advances by exactly its class's size per launch, and a published word carries the
// launch's epoch in its upper half (the host zeroes the words on first use and before the epoch could repeat).
#include <hip/hip_runtime.h>

#include "slx_device.h"
#include "slx_kernels.h"

// Timing diagnostics, never part of the product build (-DSLX_CLOUD_EXP=N; the results are WRONG): 4 = stop after the rows are in (phase 1),
// 1 = stop after the offsets are known (phases 1 + 2), 2 = do not look back at the column groups before this one.
#ifndef SLX_CLOUD_EXP
#define SLX_CLOUD_EXP 0
#endif

namespace {

typedef double vec2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool cloud_keep(double zz, double fov_min, double fov_max)
{
    return !((zz < fov_min) || (zz > fov_max));                     // the reference's `continue` test, negated
}

// A published word carries its own validity (the launch's tag in the upper half) beside the count: one relaxed 64-bit atomic store,
// one relaxed 64-bit atomic load -- no other memory depends on their order, so neither side needs a release / acquire, which at agent
// scope on this chip means writing back / invalidating a whole L2 per operation (a first version did: 118 us per cloud instead of 34).
__device__ __forceinline__ void publish(unsigned long long *word, unsigned tag, unsigned value)
{
    __hip_atomic_store(word, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Reads of published words: all of a thread's words are asked for together and re-polled together until every one carries the launch's
// tag (a loop per word would put one memory round trip behind the other).
// BOUNDED: after `limit` rounds of polls the thread gives up and the function returns false.  The progress argument at the top of this
// file rests on how the dispatcher hands out workgroups (in index order, the classes of ticket counters within one of each other, at
// least `parts` workgroups resident); where that ever fails to hold -- compute-unit masking, a partition mode, another device -- the
// symptom must be an answer, not a hung GPU: the workgroup then raises the launch's flag, writes nothing, and the host runs the frame
// again on the two-launch path (slx_point_cloud_of_depth).  A poll is a memory round trip plus s_sleep: ~1 us, so the default of
// 16 384 rounds is ~15 ms against the few microseconds a wait lasts when all is well.
template <int N>
__device__ __forceinline__ bool await_words(const unsigned long long *const (&word)[N], const bool (&want)[N], unsigned tag, unsigned (&value)[N], unsigned limit)
{
    unsigned long long w[N];
    bool ready[N];
#pragma unroll
    for (int k = 0; k < N; k++) ready[k] = !want[k], value[k] = 0u;
    if (limit == 0u) {                                              // tests only (SLX_TUNE_CLOUD_SPIN = 1): whoever needs a word gives up, deterministically
        bool none = true;
#pragma unroll
        for (int k = 0; k < N; k++) none = none && ready[k];
        return none;
    }
    for (unsigned round = 0;; round++) {
#pragma unroll
        for (int k = 0; k < N; k++)
            if (!ready[k]) w[k] = __hip_atomic_load(word[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool all = true;
#pragma unroll
        for (int k = 0; k < N; k++) {
            if (!ready[k] && (unsigned)(w[k] >> 32) == tag) {
                ready[k] = true;
                value[k] = (unsigned)w[k];
            }
            all = all && ready[k];
        }
        if (all) return true;
        if (round >= limit) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

constexpr unsigned kThreads = SLX_CLOUD_THREADS, kWaves = kThreads / 64u;

// FAST: x and y by slx_div_item_const (fu, fv checked on the host to sit inside its range) -- the same quotient bits.
// WIDE: 16-byte loads (the width is even, the map 16-byte aligned).
// CH: 64-row chunks of a part at most (4 for the usual 256-row parts: x, y of a wave's 2 columns x CH chunks live in registers).
template <bool FAST, bool WIDE, unsigned CH>
__global__ __launch_bounds__(kThreads) void slx_cloud_fused_kernel(const SlxCloudFused q)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];    // [R][16] tile, swizzled | [kWaves][192] runs
    __shared__ unsigned s_ticket, s_gave_up, col_cnt[16], col_off[16], sib[16][16], part_sum[kWaves];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (q.stamps && tid == 0 && blockIdx.x < q.stamp_items) {       // diagnostics only (slx_debug_stamps)
        q.stamps[4 * (size_t)blockIdx.x + 0] = __builtin_amdgcn_s_memtime();
        q.stamps[4 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned P = (unsigned)q.parts, R = (unsigned)q.rows_per_part, n_slots = (unsigned)q.groups * P;
    double *tile = lds, *run = lds + (size_t)R * 16u + (size_t)wave * 192u;
    unsigned long long *totals = q.words + SLX_CLOUD_COUNTERS * 16u, *cols = totals + n_slots;
    // The ticket.  ONE counter for all workgroups serialises them (a counter hands out ~88 tickets per microsecond, tools/probes/satomic.hip:
    // 600 workgroups waited 7 us for theirs in the first version).  So the workgroups are dealt into SLX_CLOUD_COUNTERS classes by their
    // index and class c numbers its members c, c + C, c + 2 C ... in the order they start: within a class a lower ticket has started
    // earlier; across classes the dispatcher, which starts workgroups in index order, keeps the classes within one of each other, so the
    // started tickets are always a prefix of all tickets -- which is all the waits below need.
    if (tid == 0) {
        const unsigned c = blockIdx.x % SLX_CLOUD_COUNTERS;
        const unsigned members = n_slots / SLX_CLOUD_COUNTERS + (c < n_slots % SLX_CLOUD_COUNTERS ? 1u : 0u);
        const unsigned j = atomicAdd(reinterpret_cast<unsigned *>(q.words + c * 16u), 1u) - q.epoch * members;
        s_ticket = c + SLX_CLOUD_COUNTERS * j;
        s_gave_up = 0u;
    }
    if (tid < 16) col_cnt[tid] = 0;
    __syncthreads();
    const unsigned slot = s_ticket, g = slot / P, p = slot - g * P, tag = q.epoch + 1u;
    const unsigned u0 = g * 16u, v0 = p * R;
    const unsigned W = (unsigned)q.W, H = (unsigned)q.H;
    const unsigned rows = v0 < H ? (H - v0 < R ? H - v0 : R) : 0u;

    // ---- 1. the part's depths: rows in, swizzled tile, kept depths counted per column
    {
        constexpr unsigned PASS = kThreads / 8u;                    // rows one pass of the workgroup's lanes covers
        const unsigned cp = tid & 7u, r0 = tid >> 3;                // column pair, first row of this lane
        const unsigned ua = u0 + 2u * cp, ub = ua + 1u;
        const bool has_a = ua < W, has_b = ub < W;
        const unsigned uca = has_a ? ua : W - 1u, ucb = has_b ? ub : W - 1u;   // clamped: every load is legal, the count masks
        unsigned na = 0, nb = 0;
        constexpr unsigned UNROLL = 4;
        for (unsigned r = r0; r < rows; r += PASS * UNROLL) {
            vec2 zz[UNROLL];
#pragma unroll
            for (unsigned k = 0; k < UNROLL; k++) {
                const unsigned rr = r + PASS * k < rows ? r + PASS * k : rows - 1u;
                const size_t base = (size_t)(v0 + rr) * W;
                if (WIDE && has_b) zz[k] = __builtin_nontemporal_load(reinterpret_cast<const vec2 *>(q.z + base + ua));
                else zz[k] = vec2{__builtin_nontemporal_load(q.z + base + uca), __builtin_nontemporal_load(q.z + base + ucb)};
            }
#pragma unroll
            for (unsigned k = 0; k < UNROLL; k++) {
                const unsigned rr = r + PASS * k;
                if (rr < rows) {
                    na += (has_a && cloud_keep(zz[k].x, q.fov_min, q.fov_max)) ? 1u : 0u;
                    nb += (has_b && cloud_keep(zz[k].y, q.fov_min, q.fov_max)) ? 1u : 0u;
                    // element (row, c) lives at tile[row][c ^ (row & 15)]: a pair (2 cp, 2 cp + 1) stays one 16-byte slot, swapped
                    // when the row is odd
                    const unsigned key = rr & 15u;
                    const unsigned at = rr * 16u + ((2u * cp) ^ (key & 14u));
                    *reinterpret_cast<vec2 *>(tile + at) = (key & 1u) ? vec2{zz[k].y, zz[k].x} : zz[k];
                }
            }
        }
        // the lanes of a wave that share a column pair (cp = lane & 7) add up first: one LDS atomic per wave and column
#pragma unroll
        for (int d = 8; d < 64; d <<= 1) {
            na += __shfl_xor(na, d);
            nb += __shfl_xor(nb, d);
        }
        if (lane < 8u) {
            atomicAdd(&col_cnt[2u * cp], na);
            atomicAdd(&col_cnt[2u * cp + 1u], nb);
        }
    }
    __syncthreads();
#if SLX_CLOUD_EXP & 4
    return;
#endif

    // ---- 2. publish this part's counts, collect the offsets
    if (tid < 16) publish(cols + (size_t)slot * 16u + tid, tag, col_cnt[tid]);
    if (tid == 0) {
        unsigned t = 0;
        for (int c = 0; c < 16; c++) t += col_cnt[c];
        publish(totals + slot, tag, t);
    }
    // ---- 3a. while the other parts publish: x and y of this wave's columns, into registers.  Nothing here needs the offsets; the
    // look-back below then mostly finds its words already tagged (the parts of a launch run in step, so without this a workgroup
    // idles for the skew of the launch -- ~6 us of a 21 us kernel in the first version)
    constexpr unsigned kCols = 16u / kWaves, kChunks = CH;          // columns a wave owns, 64-row chunks of a part at most
    const double ru = FAST ? slx_refined_rcp_f64(q.fu) : 0.0, rv = FAST ? slx_refined_rcp_f64(q.fv) : 0.0;
    // (FAST only: the literal divisions of the other instantiations -- fu or fv outside the cheap sequence's range, never a real
    // calibration -- need more registers than 5 waves per SIMD leave; they compute x and y in step 3b instead)
    double xo[FAST ? kCols : 1][FAST ? kChunks : 1], yo[FAST ? kCols : 1][FAST ? kChunks : 1];
    if (FAST && q.xyz) {
#pragma unroll
        for (unsigned ci = 0; ci < kCols; ci++) {
            const unsigned c = wave + ci * kWaves;
            const double uc = (double)(int)(u0 + c) - q.cx;         // R/CCalculation.cpp:762
#pragma unroll
            for (unsigned ch = 0; ch < kChunks; ch++) {
                const unsigned rr = ch * 64u + lane;
                const unsigned rc = rr < rows ? rr : (rows ? rows - 1u : 0u);
                const double zc = tile[rc * 16u + (c ^ (rc & 15u))];
                const double vc = (double)((int)(v0 + rr) + q.row_offset) - q.cy;               // :763
                xo[FAST ? ci : 0][FAST ? ch : 0] = slx_div_item_const(zc * uc, q.fu, ru);      // :766
                yo[FAST ? ci : 0][FAST ? ch : 0] = slx_div_item_const(zc * vc, q.fv, rv);      // :767
            }
        }
    }

    unsigned before = 0;
    {
        // this thread's words: the column count (part, column) = tid of this group's parts, and every kThreads-th total of the parts
        // of the column groups before this one, four at a time
        const unsigned n_before = g * P;
        const unsigned long long *word[4];
        bool want[4];
        word[0] = cols + (size_t)(g * P) * 16u + tid;
        want[0] = tid < P * 16u;
#if SLX_CLOUD_EXP & 2
        const unsigned first = n_before;
#else
        const unsigned first = 0;
#endif
#pragma unroll
        for (int k = 1; k < 4; k++) {
            const unsigned s = first + tid + (unsigned)(k - 1) * kThreads;
            word[k] = totals + s;
            want[k] = s < n_before;
        }
        const unsigned long long *const wd[4] = {word[0], word[1], word[2], word[3]};
        unsigned got[4];
        bool arrived = await_words(wd, want, tag, got, q.spin_limit);
        if (want[0]) sib[tid >> 4][tid & 15u] = got[0];
        before = got[1] + got[2] + got[3];
        for (unsigned s = first + tid + 3u * kThreads; arrived && s < n_before; s += kThreads) {   // maps with more than 3 x kThreads parts before this one
            const unsigned long long *const one[1] = {totals + s};
            const bool yes[1] = {true};
            unsigned v[1];
            arrived = await_words(one, yes, tag, v, q.spin_limit);
            before += v[0];
        }
        if (!arrived) s_gave_up = 1u;                               // (every writer stores the same value)
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d);
    if (lane == 0) part_sum[wave] = before;
    __syncthreads();
    if (s_gave_up) {
        // A word this part needs never came within the bound.  Its own counts are published (nobody waits for THIS workgroup in vain);
        // it writes no point and no total, and tells the host, which repeats the frame on the two-launch path.
        if (tid == 0) {
            __hip_atomic_store(q.gave_up_host, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    unsigned groups_before = 0;
#pragma unroll
    for (unsigned k = 0; k < kWaves; k++) groups_before += part_sum[k];
    if (tid < 16) {
        unsigned off = groups_before;
        for (unsigned c = 0; c < tid; c++)
            for (unsigned pp = 0; pp < P; pp++) off += sib[pp][c];
        for (unsigned pp = 0; pp < p; pp++) off += sib[pp][tid];
        col_off[tid] = off;
    }
    if (tid == 0 && g + 1u == (unsigned)q.groups && p == 0) {       // the last group knows the number of points
        unsigned total = groups_before;
        for (unsigned c = 0; c < 16; c++)
            for (unsigned pp = 0; pp < P; pp++) total += sib[pp][c];
        *q.total_dev = total;
        if (q.total_host) *q.total_host = total;
    }
    __syncthreads();
    if (!q.xyz) return;                                             // count only: the caller wanted the number of points
#if SLX_CLOUD_EXP & 1
    return;
#endif

    // ---- 3b. columns out: a wave per column, 64 rows at a time -- rank the kept depths, pack the records, store the run
#pragma unroll
    for (unsigned ci = 0; ci < kCols; ci++) {
        const unsigned c = wave + ci * kWaves;
        if (u0 + c >= W) break;                                     // uniform over the wave
        double *dst = q.xyz + (size_t)col_off[c] * 3u;
#pragma unroll
        for (unsigned ch = 0; ch < kChunks; ch++) {
            if (ch * 64u >= rows) break;                            // uniform over the workgroup
            const unsigned rr = ch * 64u + lane;
            const unsigned rc = rr < rows ? rr : rows - 1u;
            const double zc = tile[rc * 16u + (c ^ (rc & 15u))];
            const bool keep = rr < rows && cloud_keep(zc, q.fov_min, q.fov_max);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
            if (keep) {
                const unsigned rank = (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                if constexpr (FAST) {
                    run[3u * rank + 0u] = xo[ci][ch];
                    run[3u * rank + 1u] = yo[ci][ch];
                } else {
                    const double uc = (double)(int)(u0 + c) - q.cx;                              // R/CCalculation.cpp:762
                    const double vc = (double)((int)(v0 + rr) + q.row_offset) - q.cy;            // :763
                    run[3u * rank + 0u] = zc * uc / q.fu;                                        // :766
                    run[3u * rank + 1u] = zc * vc / q.fv;                                        // :767
                }
                run[3u * rank + 2u] = zc;
            }
            // the chunk's records leave as one contiguous run of doubles (a wave's LDS accesses execute in order: the reads below
            // see the writes above, and the next chunk's writes come after these reads)
            __builtin_amdgcn_wave_barrier();
            const unsigned words = 3u * (unsigned)__builtin_popcountll(m);
#pragma unroll
            for (unsigned j = 0; j < 3u; j++) {
                const unsigned w = lane + 64u * j;
                if (w < words) __builtin_nontemporal_store(run[w], dst + w);
            }
            dst += words;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (q.stamps && blockIdx.x < q.stamp_items) {                   // diagnostics: after every wave of the workgroup has issued its last store
        __syncthreads();
        if (tid == 0) {
            q.stamps[4 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
            q.stamps[4 * (size_t)blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

}  // namespace

int slx_launch_cloud_fused(const SlxCloudFused &q, void *stream)
{
    if (q.groups < 1 || q.parts < 1 || q.parts > 16 || q.rows_per_part < 1 || !q.z || !q.words || !q.total_dev) return (int)hipErrorInvalidValue;
    auto in_range = [](double d) { return __builtin_fabs(d) > 0x1p-90 && __builtin_fabs(d) < 0x1p90; };
    const bool fast = in_range(q.fu) && in_range(q.fv);
    const bool wide = (q.W % 2) == 0 && (reinterpret_cast<uintptr_t>(q.z) % 16) == 0;
    if (q.rows_per_part > SLX_CLOUD_MAX_ROWS) return (int)hipErrorInvalidValue;
    typedef void (*cloud_fn)(const SlxCloudFused);
    constexpr unsigned kTall = SLX_CLOUD_MAX_ROWS / 64u;
    static const cloud_fn table[2][2][2] = {
        {{slx_cloud_fused_kernel<false, false, 4>, slx_cloud_fused_kernel<false, false, kTall>}, {slx_cloud_fused_kernel<false, true, 4>, slx_cloud_fused_kernel<false, true, kTall>}},
        {{slx_cloud_fused_kernel<true, false, 4>, slx_cloud_fused_kernel<true, false, kTall>}, {slx_cloud_fused_kernel<true, true, 4>, slx_cloud_fused_kernel<true, true, kTall>}}};
    const cloud_fn fn = table[fast ? 1 : 0][wide ? 1 : 0][q.rows_per_part > 256 ? 1 : 0];
    const size_t lds = slx_cloud_fused_lds_bytes(q.rows_per_part);
    hipLaunchKernelGGL(fn, dim3((unsigned)(q.groups * q.parts)), dim3(SLX_CLOUD_THREADS), lds, (hipStream_t)stream, q);
    return (int)hipGetLastError();
}
