// slx_track.hip -- the dynamic-frame tracker of DynaFrame for gfx950.
//
//   CCalculation::StripRegression      R/CCalculation.cpp:789-892  -> slx_strip_regression_kernel
//   CCalculation::FillOtherDeltaProU   R/CCalculation.cpp:595-663  -> slx_delta_p_kernel + slx_track_update_kernel
//   CCalculation::FillCoordinate(fN)   R/CCalculation.cpp:666-785  -> inside slx_track_update_kernel (deltaZ: :772-775)
// cv::blur (OpenCV 2.4.9 boxFilter) is restated: normalised 3x3, BORDER_REFLECT_101, double sums scaled by 1./9.
// All sums here are sums of small integers held in floats/doubles and therefore exact in any order.
// R/ = DynaFrame/DynaFrame/ of the reference repository.
#include <hip/hip_runtime.h>

#include "slx_device.h"
#include "slx_kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int kTile = 256;            // threads per workgroup = columns per tile, halo included
constexpr int kBandRows = 8;          // rows per workgroup: 1920x1200 gives 9 x 150 workgroups, ~5 per CU

// One lane per column keeps the 21-row sliding sum of its column while the workgroup walks down a band of rows;
// every row's sums go through LDS so that a lane can scan its 20 horizontal neighbours (win/2 to the left,
// win/2 - 1 to the right, in the reference's order: the centre wins ties, then the leftmost).  The kernel writes
// every pixel of the two strip planes -- 0 outside the interior (R/CCalculation.cpp:799-802, :827-828) -- so the
// planes need no clearing pass.  HWT > 0: the half window as a compile-time constant (the scan unrolls); 0: win / 2.
template <int HWT>
__global__ __launch_bounds__(kTile) void slx_strip_regression_kernel(const uint8_t *cam, size_t stride, int W, int H, int win,
                                                                     float *stripW, float *stripB)
{
    __shared__ float row_sum[2][kTile];
    const int hw = HWT > 0 ? HWT : win / 2;
    const int out_cols = kTile - 2 * hw;
    const int tx = threadIdx.x;
    // lane tx holds column tile_first - hw + tx; lanes hw .. kTile-hw-1 own the tile's out_cols output columns
    const int c = (int)blockIdx.x * out_cols + tx - hw;
    const bool owns = tx >= hw && tx < kTile - hw && c < W;
    const bool col_interior = c >= hw && c < W - hw;               // valSum is 0 elsewhere
    const int r0 = (int)blockIdx.y * kBandRows;
    const int r1 = r0 + kBandRows < H ? r0 + kBandRows : H;
    float sum = 0.f;
    bool have = false;
    for (int h = r0; h < r1; h++) {
        if (h < hw || h >= H - hw) {                               // border row (uniform over the workgroup)
            if (owns) {
                stripW[(size_t)h * W + c] = 0.f;
                stripB[(size_t)h * W + c] = 0.f;
            }
            continue;
        }
        if (!have) {                                               // first interior row of the band: :814-819
            if (col_interior)
                for (int r = h - hw; r <= h + hw; r++) sum += (float)cam[(size_t)r * stride + c];
            have = true;
        } else if (col_interior) {                                 // :820-822
            sum = sum - (float)cam[(size_t)(h - 1 - hw) * stride + c] + (float)cam[(size_t)(h + hw) * stride + c];
        }
        float *buf = row_sum[h & 1];
        buf[tx] = col_interior ? sum : 0.f;
        __syncthreads();
        // the other LDS buffer is written next; this one is written again only two rows (one barrier) later
        if (owns) {
            float mxi = 0.f, mni = 0.f;
            if (col_interior) {
                float mx = buf[tx], mn = mx;
#pragma unroll
                for (int i = -hw; i < hw; i++) {                   // :838-851
                    const float v = buf[tx + i];
                    if (v > mx) { mx = v; mxi = (float)i; }
                    if (v < mn) { mn = v; mni = (float)i; }
                }
            }
            stripB[(size_t)h * W + c] = mni;
            stripW[(size_t)h * W + c] = mxi;
        }
    }
}

// The same for a compile-time half window HW (the reference's RECO_WINDOW_SIZE = 21 -> HW = 10), restructured so that
// nothing in it is serial: a lane loads the kBandRows + 2 HW bytes of its column that the band needs in one go, forms the
// band's sliding sums from them, and all rows' sums go to LDS behind ONE barrier; then every row is scanned with the
// neighbour's position folded into the compared key, so that the scan is v_max3_u32 / v_min3_u32 instead of
// compare + two selects per neighbour and strip:
//   white strip: key = sum * 32 + priority, priority 2 HW for the centre, HW - 1 - i for neighbour i (leftmost highest)
//                -> the maximum key is the largest sum, the centre on ties, else the leftmost        (:838-851, `>` replaces)
//   black strip: key = sum * 32 + rank, rank 0 for the centre, i + HW + 1 for neighbour i -> the minimum key likewise.
// Sums are at most 21 * 255 < 2^13, so the keys are exact 18-bit integers.
template <int HW>
__global__ __launch_bounds__(kTile) void slx_strip_regression_band_kernel(const uint8_t *cam, size_t stride, int W, int H, float *stripW, float *stripB,
                                                                          const float *prevW, const float *prevB, float *raw)
{
    static_assert(2 * HW <= 31, "the neighbour rank must fit 5 bits");
    __shared__ uint32_t sums[kBandRows][kTile];
    constexpr int out_cols = kTile - 2 * HW;
    const int tx = threadIdx.x;
    const int c = (int)blockIdx.x * out_cols + tx - HW;
    const bool owns = tx >= HW && tx < kTile - HW && c < W;
    const bool col_interior = c >= HW && c < W - HW;
    const int r0 = (int)blockIdx.y * kBandRows;
    const int r1 = r0 + kBandRows < H ? r0 + kBandRows : H;
    const int ha = r0 > HW ? r0 : HW, hb = r1 < H - HW ? r1 : H - HW;   // interior rows of the band: [ha, hb)
    if (ha < hb) {
        uint32_t b[kBandRows + 2 * HW];
#pragma unroll
        for (int k = 0; k < kBandRows + 2 * HW; k++) {
            const int r = ha - HW + k;
            b[k] = (col_interior && r < hb + HW) ? cam[(size_t)r * stride + c] : 0u;
        }
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k <= 2 * HW; k++) sum += b[k];
#pragma unroll
        for (int j = 0; j < kBandRows; j++) {
            sums[j][tx] = sum << 5;                                 // 0 outside the interior columns, as valSum is
            if (j + 1 < kBandRows) sum = sum - b[j] + b[j + 2 * HW + 1];
        }
    }
    __syncthreads();
    if (!owns) return;
#pragma unroll
    for (int j = 0; j < kBandRows; j++) {
        const int h = r0 + j;
        if (h >= r1) break;
        float mxi = 0.f, mni = 0.f;
        const int jj = h - ha;                                      // row of `sums`
        if (col_interior && h >= ha && h < hb) {
            const uint32_t *row = &sums[jj][tx];
            uint32_t kmax = row[0] | (uint32_t)(2 * HW), kmin = row[0];
#pragma unroll
            for (int i = -HW; i < HW; i += 2) {
                const uint32_t s0 = row[i], s1 = row[i + 1];
                // neighbour 0 is the centre again with a lower priority / higher rank: harmless
                const uint32_t w0 = s0 | (uint32_t)(HW - 1 - i), w1 = s1 | (uint32_t)(HW - 2 - i);
                const uint32_t k0 = s0 | (uint32_t)(i + HW + 1), k1 = s1 | (uint32_t)(i + HW + 2);
                kmax = max(max(kmax, w0), w1);
                kmin = min(min(kmin, k0), k1);
            }
            const int p = (int)(kmax & 31u), q = (int)(kmin & 31u);
            mxi = p == 2 * HW ? 0.f : (float)(HW - 1 - p);
            mni = q == 0 ? 0.f : (float)(q - HW - 1);
        }
        const size_t o = (size_t)h * W + c;
        stripB[o] = mni;
        stripW[o] = mxi;
        if (raw) {                                                  // deltaP selection of the next step, R/CCalculation.cpp:602-617
            const float dB = prevB[o] - mni, dW = prevW[o] - mxi;
            raw[o] = (__builtin_fabsf(dB) < __builtin_fabsf(dW)) ? dB : dW;
        }
    }
}

// R/CCalculation.cpp:602-617
__global__ __launch_bounds__(256) void slx_delta_p_kernel(const float *W0, const float *B0, const float *W1, const float *B1, size_t n, float *raw)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float f0W = W0[i], f0B = B0[i], f1W = W1[i], f1B = B1[i];
    const float dB = f0B - f1B, dW = f0W - f1W;
    raw[i] = (__builtin_fabsf(dB) < __builtin_fabsf(dW)) ? dB : dW;
}

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// blur 3x3 (:650), U += deltaP (:656-658), FillCoordinate(fN) (:672-708, :756-771), deltaZ (:772-775)
__global__ __launch_bounds__(256) void slx_track_update_kernel(const float *raw, float *deltaP, double *U, double *z, double *x, double *y,
                                                               double *deltaZ, const SlxKParams p)
{
    const int W = p.width, H = p.height;
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= W || v >= H) return;
    double s = 0.0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) s += (double)raw[(size_t)reflect101(v + dy, H) * W + reflect101(u + dx, W)];
    const float dp = (float)(s * (1. / 9));
    const size_t i = (size_t)v * W + u;
    deltaP[i] = dp;
    const double Uv = U[i] + (double)dp;
    U[i] = Uv;
    const double uc = (double)u - p.cx, vc = (double)(v + p.row_offset) - p.cy;
    const double cC = ((uc * p.fv) * p.P00 + (vc * p.fu) * p.P01) + p.K1;
    const double cD = ((uc * p.fv) * p.P20 + (vc * p.fu) * p.P21) + p.K2;
    double zz = -(p.cA - p.cB * Uv) / (cC - cD * Uv);
    if ((zz < p.fov_min) || (zz > p.fov_max)) zz = 0.0;
    if (Uv == 0.0) zz = 0.0;                                        // the reference leaves z untouched here; defined 0
    deltaZ[i] = zz - z[i];
    z[i] = zz;
    if (x) x[i] = zz * uc / p.fu;
    if (y) y[i] = zz * vc / p.fv;
}

// One dynamic frame in ONE launch (the reference's window, HW = 10): StripRegression(fN), the deltaP selection against the
// previous frame's strips, cv::blur 3x3, U += deltaP, FillCoordinate(fN) and deltaZ.  The unblurred deltaP never goes to
// HBM: a workgroup computes it for its band of kFusedRows rows plus one halo row above and below and one halo column either
// side (the strips of those halo pixels are computed a second time by the neighbouring workgroup: 25 % more scan work for
// 8 bytes per pixel less traffic and one launch less), keeps it in LDS, and blurs from there.
// Tile geometry (round 5): 254 of a workgroup's 256 lanes own an output column (the first and the last lane stand on the blur's halo
// columns and only scan).  The sliding sums cover the 2 HW = 20 columns the scan reaches beyond them as well, 276 in all, and are formed
// TWO columns per lane by the first 138 lanes (16-bit loads, both columns in the halves of one register).  Before, the scan's reach
// took 22 of every workgroup's lanes: 1920 columns were 9 tiles of 234, the ninth a fifth full, 1 350 workgroups for 256 CUs; now 8
// tiles of 254, 1 200 workgroups: 32.4 -> 31.3 us per frame on one box (profiles/r05_track_tile_geometry_ab.log).  Two versions
// that formed the extra sums columns, or scanned the halo columns, in a SECOND pass of a few lanes were 14 % and 25 % slower than
// the 234-column tiles: the phase before the first barrier is on every workgroup's critical path, and so is every row's scan.
//   phase 1  two columns per lane: the 30 image bytes each column contributes to the 10 rows' sliding sums -> LDS
//   phase 2  the +-10 scan of every row (max3 / min3 over keys, as in the band kernel), strips out for the pixels this workgroup
//            owns, deltaP selection -> LDS
//   phase 3  3x3 box sum from LDS (BORDER_REFLECT_101 at the image border), U, depth, x, y, deltaZ
// a7's second pass divides by the constants fu, fv: the refined reciprocal is formed once, the quotient takes the residual
// correction of the IEEE sequence, and anything that sequence cannot do unscaled (zero / NaN) goes to the literal division.
constexpr int kFusedRows = 8;

template <int HW>
__global__ __launch_bounds__(kTile) void slx_track_fused_kernel(const uint8_t *cam, size_t stride, float *stripW, float *stripB, const float *prevW,
                                                                const float *prevB, float *deltaP, double *U, double *z, double *x, double *y,
                                                                double *deltaZ, const SlxKParams p, unsigned tiles_x)
{
    static_assert(2 * HW <= 31, "the neighbour rank must fit 5 bits");
    constexpr int OUT = kTile - 2;                                   // output columns per workgroup: every lane but the first and the last
    constexpr int RR = kFusedRows + 2;                               // rows of unblurred deltaP a band needs
    constexpr int XS = 2 * HW;                                       // extra columns of sliding sums: the scan's reach, both sides
    constexpr int SW = kTile + XS;                                   // sums column si holds image column c0 - 1 - HW + si
    static_assert(SW % 2 == 0, "two columns of sums per lane");
    __shared__ __attribute__((aligned(8))) uint32_t sums[RR][SW];
    __shared__ float rawt[RR][kTile];                                // rawt column tx holds image column c0 - 1 + tx: lane tx's own
    const int W = p.width, H = p.height;
    const int tx = threadIdx.x;
    // diagnostics only (slx_debug_stamps): when this workgroup started and ended, by the shader clock and the 100 MHz real-time clock
    if (p.stamps && tx == 0 && blockIdx.x < p.stamp_items) {
        p.stamps[4 * (size_t)blockIdx.x + 0] = __builtin_amdgcn_s_memtime();
        p.stamps[4 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    }
    // XCD-aware tile order (round 5).  The dispatcher deals workgroups round-robin over the 8 XCDs, each with its own L2; a band shares 22
    // of its 30 image rows and 2 of its 10 rows of the previous frame's strips with the bands above and below it, and in plain order
    // those neighbours sit on other XCDs: the shared rows came from HBM once per band (74.9 MB read per frame for 57.6 algorithmic,
    // profiles/r05_track_pmc_summary.json).  Here XCD x takes a contiguous run of tiles in row-major order, so that a tile's vertical
    // neighbours run on its own XCD at about the same time and the shared rows are L2 hits (58.6 MB).
    // Placement only affects speed; any order gives the same result.
    unsigned tile = blockIdx.x;
    {
        const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = tile & 7u, within = tile >> 3;
        tile = xcd * q + (xcd < r ? xcd : r) + within;
    }
    const int tile_x = (int)(tile % tiles_x), tile_y = (int)(tile / tiles_x);
    const int c0 = tile_x * OUT, c = c0 - 1 + tx;                    // image column of this lane; lanes 0 and kTile - 1 are the blur's halo columns
    const int r0 = tile_y * kFusedRows;
    const int r1 = r0 + kFusedRows < H ? r0 + kFusedRows : H;
    const int ra = r0 - 1;                                           // image row of tile row 0
    const int ha = ra > HW ? ra : HW, hb = ra + RR < H - HW ? ra + RR : H - HW;   // interior rows of the tile: [ha, hb)
    const bool col_in = c >= 0 && c < W;
    const bool owns_col = tx >= 1 && tx < kTile - 1 && c < W;
    // Everything a workgroup reads besides the image is asked for up front -- the previous frame's strips for all ten rows here,
    // U and z two rows ahead of their use below -- so that those latencies pass behind the sums and the scans: all workgroups of
    // a frame are resident at once, and a workgroup's own dependent chain is what the launch lasts.
    float pB[RR], pW[RR];
#pragma unroll
    for (int j = 0; j < RR; j++) {
        const int h = ra + j;
        const bool need = col_in && h >= 0 && h < H;
        const size_t o = (size_t)(need ? h : 0) * W + (need ? c : 0);
        pB[j] = need ? prevB[o] : 0.f;
        pW[j] = need ? prevW[o] : 0.f;
    }
    double Uq[kFusedRows + 1], zq[kFusedRows + 1];
    auto fetch_row = [&](int j) {                                    // inputs of output row j (tile row j, image row ra + j)
        const int v = ra + j;
        const bool need = owns_col && v < r1;
        const size_t i = (size_t)(need ? v : 0) * W + (need ? c : 0);
        Uq[j] = need ? U[i] : 0.0;
        zq[j] = need ? z[i] : 0.0;
    };
    fetch_row(1);
    fetch_row(2);
    fetch_row(3);
    // Sliding sums, TWO image columns per lane: lanes 0 .. SW/2 - 1 load the 30 rows of columns c0 - 1 - HW + 2 tx and + 1 as one
    // 16-bit word each (any alignment), keep the two bytes in the halves of a register (a sum of 21 bytes fits 16 bits, and a sliding
    // sum never borrows: what is subtracted was added before) and run ONE sliding sum over both; the last wave has no part in this phase.
    if (ha < hb && tx < SW / 2) {
        const int cs = c0 - 1 - HW + 2 * tx;                         // the even one of this lane's two columns
        const uint32_t keep = ((cs >= HW && cs < W - HW) ? 0xffffu : 0u) | ((cs + 1 >= HW && cs + 1 < W - HW) ? 0xffff0000u : 0u);   // valSum is 0 elsewhere
        const int cl = cs < 0 ? 0 : cs > W - 2 ? W - 2 : cs;         // a pair that had to move holds no interior column (HW >= 1)
        uint32_t b[RR + 2 * HW];
#pragma unroll
        for (int k = 0; k < RR + 2 * HW; k++) {
            const int r = ha - HW + k;
            uint16_t two = 0;
            if (r < hb + HW) __builtin_memcpy(&two, cam + (size_t)r * stride + cl, 2);
            b[k] = ((uint32_t)two | ((uint32_t)two << 8)) & 0x00ff00ffu;
        }
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k <= 2 * HW; k++) sum += b[k];
#pragma unroll
        for (int j = 0; j < RR; j++) {
            const uint32_t m = sum & keep;
            *reinterpret_cast<uint2 *>(&sums[j][2 * tx]) = make_uint2((m & 0xffffu) << 5, (m >> 16) << 5);   // row ha + j
            if (j + 1 < RR) sum = sum - b[j] + b[j + 2 * HW + 1];
        }
    }
    __syncthreads();
    // strips and deltaP selection of this lane's column at tile row j (image row ra + j) -> rawt[j][tx]; strips out when owned
    auto scan_row = [&](int j) {
        const int h = ra + j;
        float r = 0.f;
        if (col_in && h >= 0 && h < H) {
            float mxi = 0.f, mni = 0.f;
            if (c >= HW && c < W - HW && h >= ha && h < hb) {
                const uint32_t *row = &sums[h - ha][tx + HW];
                uint32_t kmax = row[0] | (uint32_t)(2 * HW), kmin = row[0];
#pragma unroll
                for (int i = -HW; i < HW; i += 2) {
                    const uint32_t s0 = row[i], s1 = row[i + 1];
                    const uint32_t w0 = s0 | (uint32_t)(HW - 1 - i), w1 = s1 | (uint32_t)(HW - 2 - i);
                    const uint32_t k0 = s0 | (uint32_t)(i + HW + 1), k1 = s1 | (uint32_t)(i + HW + 2);
                    kmax = max(max(kmax, w0), w1);
                    kmin = min(min(kmin, k0), k1);
                }
                const int pw = (int)(kmax & 31u), q = (int)(kmin & 31u);
                mxi = pw == 2 * HW ? 0.f : (float)(HW - 1 - pw);
                mni = q == 0 ? 0.f : (float)(q - HW - 1);
            }
            if (owns_col && h >= r0 && h < r1) {
                const size_t o = (size_t)h * W + c;
                stripB[o] = mni;
                stripW[o] = mxi;
            }
            const float dB = pB[j] - mni, dW = pW[j] - mxi;          // R/CCalculation.cpp:602-617
            r = (__builtin_fabsf(dB) < __builtin_fabsf(dW)) ? dB : dW;
        }
        rawt[j][tx] = r;
    };
    const double uc = (double)c - p.cx;
    const double aC = (uc * p.fv) * p.P00, aD = (uc * p.fv) * p.P20;
    const double rfu = slx_refined_rcp_f64(p.fu), rfv = slx_refined_rcp_f64(p.fv);
    const int dl = c - 1 < 0 ? 1 : -1, dr = c + 1 >= W ? -1 : 1;     // BORDER_REFLECT_101: column -1 is column 1, column W is column W-2
    const int ri = owns_col ? tx : 1;                                // this lane's column of rawt (the halo lanes read nothing)
    // Row by row: the scan of row j+1 and the output of row j alternate, so that a workgroup's loads and stores are spread
    // over its lifetime instead of coming in one burst at the end (every workgroup of a frame is resident at once and they all
    // run the same timeline: phases would otherwise line up across the whole chip).
    scan_row(0);
    scan_row(1);
#pragma unroll
    for (int j = 1; j <= kFusedRows; j++) {
        const int v = ra + j;                                        // uniform over the workgroup
        if (v >= r1) break;
        const size_t i = (size_t)v * W + (owns_col ? c : 0);
        if (j + 3 <= kFusedRows) fetch_row(j + 3);                   // three rows ahead
        const double Uin = Uq[j], zin = zq[j];
        scan_row(j + 1);
        __syncthreads();                                             // rows j-1 .. j+1 of rawt are complete; no row is ever rewritten
        if (!owns_col) continue;
        const int ju = v - 1 < 0 ? j + 1 : j - 1, jd = v + 1 >= H ? j - 1 : j + 1;
        double s = 0.0;                                              // sums of small integers: exact in any order
        for (int jj : {ju, j, jd}) s += ((double)rawt[jj][ri + dl] + (double)rawt[jj][ri]) + (double)rawt[jj][ri + dr];
        const float dp = (float)(s * (1. / 9));                     // cv::blur: the box sum times 1./9 (:650)
        deltaP[i] = dp;
        const double Uv = Uin + (double)dp;                         // :656-658
        U[i] = Uv;
        const double vc = (double)(v + p.row_offset) - p.cy;
        const double cC = (aC + (vc * p.fu) * p.P01) + p.K1;
        const double cD = (aD + (vc * p.fu) * p.P21) + p.K2;
        double zz = -(p.cA - p.cB * Uv) / (cC - cD * Uv);
        if ((zz < p.fov_min) || (zz > p.fov_max)) zz = 0.0;
        if (Uv == 0.0) zz = 0.0;                                     // the reference leaves z untouched here; defined 0
        deltaZ[i] = zz - zin;                                        // :772-775
        z[i] = zz;
        if (x) x[i] = slx_div_item_const(zz * uc, p.fu, rfu);        // :766
        if (y) y[i] = slx_div_item_const(zz * vc, p.fv, rfv);        // :767
    }
    if (p.stamps && blockIdx.x < p.stamp_items) {                    // diagnostics: after every wave of the workgroup has issued its last store
        __syncthreads();
        if (tx == 0) {
            p.stamps[4 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
            p.stamps[4 * (size_t)blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

}  // namespace

static int slx_launch_strip_regression_only(const uint8_t *cam, size_t stride, int W, int H, int win, float *stripW, float *stripB, void *stream,
                                            const float *prevW, const float *prevB, float *raw);

int slx_launch_strip_regression(const uint8_t *cam, size_t stride, int W, int H, int win, float *stripW, float *stripB, void *stream,
                                const float *prevW, const float *prevB, float *raw)
{
    const bool fuse = raw && win / 2 == 10 && H > 20 && W > 20;      // the band kernel also forms deltaP's raw selection
    int e = slx_launch_strip_regression_only(cam, stride, W, H, win, stripW, stripB, stream, fuse ? prevW : nullptr, fuse ? prevB : nullptr,
                                             fuse ? raw : nullptr);
    if (e == 0 && raw && !fuse) e = slx_launch_delta_p(prevW, prevB, stripW, stripB, (size_t)W * H, raw, stream);
    return e;
}

static int slx_launch_strip_regression_only(const uint8_t *cam, size_t stride, int W, int H, int win, float *stripW, float *stripB, void *stream,
                                            const float *prevW, const float *prevB, float *raw)
{
    const int hw = win / 2;
    if (win < 3 || 2 * hw >= kTile - 1 || H <= 2 * hw || W <= 2 * hw) {              // no interior: the strips are 0
        hipError_t e = hipMemsetAsync(stripW, 0, (size_t)W * H * sizeof(float), (hipStream_t)stream);
        if (e == hipSuccess) e = hipMemsetAsync(stripB, 0, (size_t)W * H * sizeof(float), (hipStream_t)stream);
        return (int)e;
    }
    const int out_cols = kTile - 2 * hw;
    const dim3 grid((unsigned)((W + out_cols - 1) / out_cols), (unsigned)((H + kBandRows - 1) / kBandRows));
    if (hw == 10)                                                                      // RECO_WINDOW_SIZE = 21, R/StaticParameters.cpp
        hipLaunchKernelGGL(slx_strip_regression_band_kernel<10>, grid, dim3(kTile), 0, (hipStream_t)stream, cam, stride, W, H, stripW, stripB, prevW,
                           prevB, raw);
    else
        hipLaunchKernelGGL(slx_strip_regression_kernel<0>, grid, dim3(kTile), 0, (hipStream_t)stream, cam, stride, W, H, win, stripW, stripB);
    return (int)hipGetLastError();
}

int slx_launch_delta_p(const float *W0, const float *B0, const float *W1, const float *B1, size_t n, float *raw, void *stream)
{
    hipLaunchKernelGGL(slx_delta_p_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W0, B0, W1, B1, n, raw);
    return (int)hipGetLastError();
}

int slx_launch_track_update(const SlxKParams &kp, const float *raw, float *deltaP, double *U, double *z, double *x, double *y, double *deltaZ,
                            void *stream)
{
    const dim3 grid((unsigned)((kp.width + 63) / 64), (unsigned)((kp.height + 3) / 4));
    hipLaunchKernelGGL(slx_track_update_kernel, grid, dim3(256), 0, (hipStream_t)stream, raw, deltaP, U, z, x, y, deltaZ, kp);
    return (int)hipGetLastError();
}

bool slx_track_fusable(int W, int H, int win) { return win / 2 == 10 && (win & 1) && H > 20 && W > 20; }

int slx_launch_track_fused(const SlxKParams &kp, const uint8_t *cam, size_t stride, float *stripW, float *stripB, const float *prevW, const float *prevB,
                           float *deltaP, double *U, double *z, double *x, double *y, double *deltaZ, void *stream)
{
    constexpr int out_cols = kTile - 2;                              // every lane but the two halo lanes owns an output column (slx_track_fused_kernel)
    const unsigned tiles_x = (unsigned)((kp.width + out_cols - 1) / out_cols), tiles_y = (unsigned)((kp.height + kFusedRows - 1) / kFusedRows);
    if ((unsigned long long)tiles_x * tiles_y >= (1ull << 31)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(slx_track_fused_kernel<10>, dim3(tiles_x * tiles_y), dim3(kTile), 0, (hipStream_t)stream, cam, stride, stripW, stripB, prevW, prevB,
                       deltaP, U, z, x, y, deltaZ, kp, tiles_x);
    return (int)hipGetLastError();
}
