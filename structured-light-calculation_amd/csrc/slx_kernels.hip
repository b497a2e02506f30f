// slx_kernels.hip -- fused per-pixel structured-light decode for gfx950 (CDNA4).
//
// One kernel replaces the reference's chain of full-frame passes
//   CDecodeGray::Grey2Bin/CountResult      R/CDecodeGray.cpp:150-204   (a3, a4)
//   CDecodePhase::CountResult              R/CDecodePhase.cpp:48-80    (a1, + cvFastArctan a2)
//   the Gray/phase merge                   R/CCalculation.cpp:561-589  (a5)
//   the cC/cD tables of Init               R/CCalculation.cpp:153-166  (a6, recomputed in-kernel)
//   FillCoordinate                         R/CCalculation.cpp:666-785  (a7)
// plus the BUILD-DEFINED N-step (x1), multi-frequency unwrap (x2) and Gray mask (x3).
// Every 8-bit input plane is read once, the f64 depth map is written once, and nothing
// else touches HBM.  All arithmetic that the reference states in float/double is done with
// the same IEEE operations in the same order (no FMA contraction); the only shortcuts are
// ones that are provably bit-identical (see DESIGN.md "Exactness") and are verified
// exhaustively against the oracle by tests/test_gpu_parity.py.
//
// R/ = DynaFrame/DynaFrame/ of the reference repository.
#include <hip/hip_runtime.h>

#include "slx_kernels.h"

#pragma clang fp contract(off)

namespace {

// ---- cv::fastAtan2 constants (OpenCV 2.4.9 mathfuncs.cpp), float products in float ------
constexpr float kDeg = (float)(180.0 / 3.1415926535897932384626433832795);
constexpr float kP1 = 0.9997878412794807f * kDeg;
constexpr float kP3 = -0.3258083974640975f * kDeg;
constexpr float kP5 = 0.1555786518463281f * kDeg;
constexpr float kP7 = -0.04432655554792128f * kDeg;
constexpr float kEps = (float)2.2204460492503131e-16;   // (float)DBL_EPSILON
constexpr float kInv360 = 1.0f / 360.0f;

__device__ __forceinline__ float ubyte(uint32_t w, int j) { return (float)((w >> (8 * j)) & 0xffu); }
__device__ __forceinline__ int ibyte(uint32_t w, int j) { return (int)((w >> (8 * j)) & 0xffu); }

// a1 + a2 for N == 4, all in f32.
// Identities used (each argued in DESIGN.md, each covered by the exhaustive 511x511 test):
//  * sinValue/cosValue = (g0-g2)/2, (g1-g3)/2 enter cvFastArctan only through their signs
//    and the ratio min/max, and (a/2)/(b/2) == a/b exactly, so the halving is dropped;
//  * for integer 0 <= a <= b <= 255, b >= 1, RN(a/b) == fma(fma(-b,q0,a), r, q0) with
//    r = v_rcp_f32(b), q0 = a*r (the quotient is never within 1/510 ulp of a rounding tie);
//    ax + (float)DBL_EPSILON == ax for ax >= 0.5, and 0/(0+eps) == 0/1;
//  * RN(x/360) by the same residual correction with r = RN(1/360);
//  * (float)((double)q * (double)T) == q*T in f32 (the double product is exact), and
//    (float)((double)pix + 0.5) == pix + 0.5f (the double sum is exact for pix = 0 or >= 2^-11).
// First-octant angle for 0 <= mn <= mx (integers <= 255 held in floats), mx1 = max(mx, 1).
__device__ __forceinline__ float octant_angle(float mn, float mx1)
{
    const float r = __builtin_amdgcn_rcpf(mx1);
    const float q0 = mn * r;
    const float c = __builtin_fmaf(__builtin_fmaf(-mx1, q0, mn), r, q0);
    const float cc = c * c;
    return (((kP7 * cc + kP5) * cc + kP3) * cc + kP1) * c;
}

// LUT: the first-octant angle comes from a table in LDS (filled by slx_atan_lut_init_kernel with
// octant_angle itself) instead of being recomputed: 12 fewer VALU slots per evaluation.
template <bool LUT>
__device__ __forceinline__ float wrapped_pix_4step(float g0, float g1, float g2, float g3, float Tf,
                                                   const float *lds_tab = nullptr)
{
    const float s2 = g0 - g2;
    const float c2 = g1 - g3;
    const float as = __builtin_fabsf(s2), ac = __builtin_fabsf(c2);
    float a;
    if constexpr (LUT) {
        const float mx = __builtin_fmaxf(as, ac), mn = __builtin_fminf(as, ac);
        // entry mx(mx+1)/2 + mn, exact in f32 (< 2^24)
        const unsigned idx = (unsigned)__builtin_fmaf(__builtin_fmaf(mx, mx, mx), 0.5f, mn);
        a = lds_tab[idx];
    } else {
        a = octant_angle(__builtin_fminf(as, ac), __builtin_fmaxf(__builtin_fmaxf(as, ac), 1.0f));
    }
    a = (as > ac) ? 90.f - a : a;
    a = (c2 < 0.f) ? 180.f - a : a;
    a = (s2 < 0.f) ? 360.f - a : a;
    const float d0 = a * kInv360;
    const float d = __builtin_fmaf(__builtin_fmaf(-360.f, d0, a), kInv360, d0);
    float pix = d * Tf;
    pix = pix + 0.5f;
    pix = (pix > Tf) ? pix - Tf : pix;
    return pix;
}

// a2 literally (any float inputs): used by the x1 path, N != 4.
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float a;
    if (ax >= ay) {
        const float c = ay / (ax + kEps);
        const float cc = c * c;
        a = (((kP7 * cc + kP5) * cc + kP3) * cc + kP1) * c;
    } else {
        const float c = ax / (ay + kEps);
        const float cc = c * c;
        a = 90.f - (((kP7 * cc + kP5) * cc + kP3) * cc + kP1) * c;
    }
    if (x < 0.f) a = 180.f - a;
    if (y < 0.f) a = 360.f - a;
    return a;
}

// R/CDecodePhase.cpp:67-75 cast by cast.
__device__ __forceinline__ float pix_tail_literal(float sinValue, float cosValue, int T)
{
    const float x = fast_atan2_deg(sinValue, cosValue);
    float pix = (float)((double)(x / 360.f) * (double)T);
    pix = (float)((double)pix + 0.5);
    if (pix > (float)T) pix = pix - (float)T;
    return pix;
}

// IEEE-754 correctly rounded num/den without the range scaling and special-case fix-up of the
// general f64 division: the same v_rcp_f64 + two Newton steps + residual correction hipcc emits,
// so the quotient is bit-identical whenever no scaling would have happened.  Callers guarantee
// 2^-200 <= |den| <= 2^200 and |num| in {0} U [2^-200, 2^200] (see tri_depth).
__device__ __forceinline__ double div_f64_inrange(double num, double den)
{
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
    const double q = num * r;
    return __builtin_fma(__builtin_fma(-den, q, num), r, q);
}

// a7 for one pixel: z = -(cA - cB U)/(cC - cD U), FOV clamp, U == 0 / mask -> 0.
template <bool LEAN>
__device__ __forceinline__ double tri_depth(double Uv, double cC, double cD, double cA, double cB,
                                            double fov_min, double fov_max, bool valid)
{
    const double num = cA - cB * Uv;
    const double den = cC - cD * Uv;
    double zz;
    if constexpr (LEAN) {
        // |num|, |den| <= 2^200 is guaranteed by the host (calibration magnitudes are checked);
        // tiny or zero operands take the general division
        const bool safe = __builtin_fabs(den) >= 0x1p-200 && (num == 0.0 || __builtin_fabs(num) >= 0x1p-200);
        if (__builtin_expect(safe, 1)) zz = -div_f64_inrange(num, den);
        else zz = -num / den;
    } else {
        zz = -num / den;
    }
    if ((zz < fov_min) || (zz > fov_max)) zz = 0.0;
    if (Uv == 0.0 || !valid) zz = 0.0;
    return zz;
}

// x2 for one pixel and one stage: k = (int)floor((Uprev - pf)/T + 0.5), U = pf + k*T.
// FASTK: d = Uprev - pf is exact and a multiple of 2^-24 (every pix is), so the real value
// d/T + 0.5 is either an integer or at least 2^-24/T away from one, while both the oracle's
// rounded division and fma(d, 1/T, 0.5 + 2^-30/T) stay within 2^-35/T of it (T <= 2^14): the
// biased fma lands on the same side of every integer as the exact value, ties included.
template <bool FASTK>
__device__ __forceinline__ double unwrap_stage(double Uprev, double pf, int T, double invT, double half_biased, int &k_out)
{
    const double d = Uprev - pf;
    double kd;
    if constexpr (FASTK) kd = __builtin_floor(__builtin_fma(d, invT, half_biased));
    else kd = __builtin_floor(d / (double)T + 0.5);
    k_out = (int)kd;
    return __builtin_fma(kd, (double)T, pf);        // exact: |k*T| and pf share a 2^-24 grid below 2^53
}

// One dword = four horizontally adjacent pixels of one 8-bit plane.
__device__ __forceinline__ uint32_t load_quad(const uint8_t *plane, size_t off, bool aligned, int npx)
{
    if (aligned)
        return *reinterpret_cast<const uint32_t *>(plane + off);
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < SLX_QUAD; j++)
        if (j < npx) w |= (uint32_t)plane[off + j] << (8 * j);
    return w;
}

template <typename T>
__device__ __forceinline__ void store_quad(T *dst, size_t idx, const T (&v)[SLX_QUAD], bool aligned, int npx)
{
    if (aligned) {
        if constexpr (sizeof(T) == 8) {
            typedef T vec2 __attribute__((ext_vector_type(2)));
            vec2 *d = reinterpret_cast<vec2 *>(dst + idx);
            d[0] = vec2{v[0], v[1]};
            d[1] = vec2{v[2], v[3]};
        } else if constexpr (sizeof(T) == 4) {
            typedef T vec4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<vec4 *>(dst + idx) = vec4{v[0], v[1], v[2], v[3]};
        } else {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < SLX_QUAD; j++) w |= (uint32_t)(uint8_t)v[j] << (8 * j);
            *reinterpret_cast<uint32_t *>(dst + idx) = w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SLX_QUAD; j++)
            if (j < npx) dst[idx + j] = v[j];
    }
}

// MODE: enum slx_mode.  F: frequencies (compile-time).  N4: the reference's 4-step path.
// AUX: also write the optional outputs whose pointers are non-null.
template <int MODE, int F, bool N4, bool AUX>
__global__ __launch_bounds__(256) void slx_fused_kernel(const SlxKParams p)
{
    constexpr bool HAS_PHASE = MODE != SLX_MODE_GRAY_ONLY;
    constexpr bool HAS_GRAY = MODE == SLX_MODE_GRAY_ONLY || MODE == SLX_MODE_GRAY_PHASE ||
                              MODE == SLX_MODE_MULTIFREQ_GRAYMASK;
    constexpr bool HAS_DEPTH = MODE >= SLX_MODE_GRAY_PHASE;
    constexpr bool MASKED = MODE == SLX_MODE_MULTIFREQ_GRAYMASK;

    const unsigned set = blockIdx.y;
    const unsigned lane = threadIdx.x & 63u;
    unsigned q;
    if constexpr (MASKED) {
        // every wave owns 62 quads plus one halo quad on either side (lanes 0 and 63),
        // so the 3-tap horizontal AND of x3 never leaves the wave
        const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        q = wave * 62u + lane - 1u;
    } else {
        q = blockIdx.x * blockDim.x + threadIdx.x;
    }
    const bool active = q < p.n_quads;
    if constexpr (!MASKED) {
        if (!active) return;
    }
    const unsigned qc = active ? q : 0u;
    const unsigned vrow = qc / p.quads_per_row;
    const unsigned u0 = (qc - vrow * p.quads_per_row) * SLX_QUAD;
    const int W = p.width;
    const int npx = (int)u0 + SLX_QUAD <= W ? SLX_QUAD : W - (int)u0;
    const bool aligned = p.aligned != 0;
    const size_t in_off = (size_t)vrow * p.row_stride + u0;

    float pix[F > 0 ? F : 1][SLX_QUAD];
    int bin[SLX_QUAD];

    if constexpr (HAS_PHASE) {
        const size_t set_off = (size_t)set * p.phase_set_stride + in_off;
#pragma unroll
        for (int f = 0; f < F; f++) {
            if constexpr (N4) {
                const uint32_t w0 = load_quad(p.phase[f * 4 + 0], set_off, aligned, npx);
                const uint32_t w1 = load_quad(p.phase[f * 4 + 1], set_off, aligned, npx);
                const uint32_t w2 = load_quad(p.phase[f * 4 + 2], set_off, aligned, npx);
                const uint32_t w3 = load_quad(p.phase[f * 4 + 3], set_off, aligned, npx);
                const float Tf = (float)p.period[f];
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++)
                    pix[f][j] = wrapped_pix_4step<false>(ubyte(w0, j), ubyte(w1, j), ubyte(w2, j), ubyte(w3, j), Tf);
            } else {
                float sy[SLX_QUAD] = {0.f, 0.f, 0.f, 0.f}, sx[SLX_QUAD] = {0.f, 0.f, 0.f, 0.f};
                const int N = p.n_steps;
                for (int k = 0; k < N; k++) {
                    const uint32_t w = load_quad(p.phase[f * N + k], set_off, aligned, npx);
                    const float wy = p.wy[k], wx = p.wx[k];
#pragma unroll
                    for (int j = 0; j < SLX_QUAD; j++) {
                        const float g = ubyte(w, j);
                        sy[j] = sy[j] + g * wy;
                        sx[j] = sx[j] + g * wx;
                    }
                }
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++)
                    pix[f][j] = pix_tail_literal(sy[j] * p.wscale, sx[j] * p.wscale, p.period[f]);
            }
        }
    }

    if constexpr (HAS_GRAY) {
        const size_t set_off = (size_t)set * p.gray_set_stride + in_off;
        unsigned code[SLX_QUAD] = {0u, 0u, 0u, 0u};
        const int G = p.gray_bits;
        for (int b = 0; b < G; b++) {
            const uint32_t wa = load_quad(p.gray[2 * b], set_off, aligned, npx);
            const uint32_t wb = load_quad(p.gray[2 * b + 1], set_off, aligned, npx);
#pragma unroll
            for (int j = 0; j < SLX_QUAD; j++)      // a3: saturating (pattern - inverse) > 0
                code[j] |= (ibyte(wa, j) > ibyte(wb, j) ? 1u : 0u) << b;
        }
#pragma unroll
        for (int j = 0; j < SLX_QUAD; j++)          // a4: lut[gray] = bin
            bin[j] = (int)p.lut[code[j]];
    }

    const size_t HW = p.out_set_stride;
    const size_t oidx = (size_t)vrow * (size_t)W + u0;
    bool writer = active;
    if constexpr (MASKED) writer = active && lane >= 1u && lane <= 62u;

    if constexpr (MODE == SLX_MODE_PHASE_ONLY) {
        double o[SLX_QUAD];
#pragma unroll
        for (int j = 0; j < SLX_QUAD; j++) o[j] = (double)pix[0][j];
        store_quad(p.pix + (size_t)set * HW, oidx, o, aligned, npx);
        return;
    }
    if constexpr (MODE == SLX_MODE_GRAY_ONLY) {
        double o[SLX_QUAD];
#pragma unroll
        for (int j = 0; j < SLX_QUAD; j++) o[j] = (double)bin[j] * (double)p.gray_stripe;
        store_quad(p.gray_out + (size_t)set * HW, oidx, o, aligned, npx);
        return;
    }

    if constexpr (HAS_DEPTH) {
        double U[SLX_QUAD], grayv[SLX_QUAD];
        int kf[F > 1 ? F - 1 : 1][SLX_QUAD];
        bool valid[SLX_QUAD] = {true, true, true, true};
        const double Sd = (double)p.gray_stripe;

        if constexpr (HAS_GRAY) {
#pragma unroll
            for (int j = 0; j < SLX_QUAD; j++) grayv[j] = (double)bin[j] * Sd;
        }

        if constexpr (MODE == SLX_MODE_GRAY_PHASE) {
            // a5; (int)(gray/S) % 2 == 0  <=>  bin even, since gray == bin*S exactly
            const double Td = (double)p.period[0];
#pragma unroll
            for (int j = 0; j < SLX_QUAD; j++) {
                const double phaseVal = (double)pix[0][j];
                double ph = phaseVal;
                if ((bin[j] & 1) == 0) {
                    if (phaseVal > Td * 0.75) ph = phaseVal - Td;
                } else {
                    if (phaseVal < Td * 0.25) ph = phaseVal + Td;
                    ph = ph - 0.5 * Td;
                }
                U[j] = grayv[j] + ph;
            }
        } else {
            // x2
#pragma unroll
            for (int j = 0; j < SLX_QUAD; j++) {
                double Uf = (double)pix[0][j];
#pragma unroll
                for (int f = 1; f < F; f++) {
                    const double pf = (double)pix[f][j];
                    const int k = (int)__builtin_floor((Uf - pf) / (double)p.period[f] + 0.5);
                    Uf = pf + (double)(k * p.period[f]);
                    kf[f - 1][j] = k;
                }
                U[j] = Uf;
            }
        }

        if constexpr (MASKED) {
            // x3: stripe agreement, then a 3-tap horizontal AND through wave shuffles
            int v0[SLX_QUAD];
#pragma unroll
            for (int j = 0; j < SLX_QUAD; j++) {
                const bool ok = __builtin_fabs(U[j] - (grayv[j] + Sd * 0.5)) <= Sd;
                v0[j] = (!active || j >= npx || ok) ? 1 : 0;
            }
            const int left = __shfl_up(v0[SLX_QUAD - 1], 1);   // pixel u0-1 lives in lane-1
            const int right = __shfl_down(v0[0], 1);           // pixel u0+4 lives in lane+1
#pragma unroll
            for (int j = 0; j < SLX_QUAD; j++) {
                const int u = (int)u0 + j;
                int ok = v0[j];
                const int l = j == 0 ? left : v0[j > 0 ? j - 1 : 0];
                const int r = j == SLX_QUAD - 1 ? right : v0[j < SLX_QUAD - 1 ? j + 1 : 0];
                if (u > 0) ok &= l;
                if (u + 1 < W) ok &= r;
                valid[j] = ok != 0;
            }
        }

        if (!writer) return;

        // a6 + a7: cC, cD recomputed from 12 scalars in the reference's operation order
        const double vc = (double)((int)vrow + p.row_offset) - p.cy;
        const double tvC = (vc * p.fu) * p.P01;
        const double tvD = (vc * p.fu) * p.P21;
        double z[SLX_QUAD], xo[SLX_QUAD], yo[SLX_QUAD];
#pragma unroll
        for (int j = 0; j < SLX_QUAD; j++) {
            const double uc = (double)((int)u0 + j) - p.cx;
            const double a = uc * p.fv;
            const double cC = (a * p.P00 + tvC) + p.K1;
            const double cD = (a * p.P20 + tvD) + p.K2;
            const double Uv = U[j];
            double zz = -(p.cA - p.cB * Uv) / (cC - cD * Uv);
            if ((zz < p.fov_min) || (zz > p.fov_max)) zz = 0.0;
            if (Uv == 0.0 || !valid[j]) zz = 0.0;
            z[j] = zz;
            if constexpr (AUX) {
                xo[j] = zz * uc / p.fu;
                yo[j] = zz * vc / p.fv;
            }
        }
        store_quad(p.z + (size_t)set * HW, oidx, z, aligned, npx);

        if constexpr (AUX) {
            if (p.x) store_quad(p.x + (size_t)set * HW, oidx, xo, aligned, npx);
            if (p.y) store_quad(p.y + (size_t)set * HW, oidx, yo, aligned, npx);
            if (p.U) store_quad(p.U + (size_t)set * HW, oidx, U, aligned, npx);
            if (p.pix) {
#pragma unroll
                for (int f = 0; f < F; f++) {
                    double o[SLX_QUAD];
#pragma unroll
                    for (int j = 0; j < SLX_QUAD; j++) o[j] = (double)pix[f][j];
                    store_quad(p.pix + ((size_t)set * F + f) * HW, oidx, o, aligned, npx);
                }
            }
            if constexpr (HAS_GRAY) {
                if (p.gray_out) store_quad(p.gray_out + (size_t)set * HW, oidx, grayv, aligned, npx);
            }
            if constexpr (F > 1) {
                if (p.k) {
#pragma unroll
                    for (int f = 0; f + 1 < F; f++)
                        store_quad(p.k + ((size_t)set * (F - 1) + f) * HW, oidx, kf[f], aligned, npx);
                }
            }
            if (p.mask) {
                uint8_t m[SLX_QUAD];
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++) m[j] = valid[j] ? 1 : 0;
                store_quad(p.mask + (size_t)set * HW, oidx, m, aligned, npx);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// Fast path: persistent workgroups walking column strips.
//
// A thread owns one quad column (4 adjacent pixels) and walks down `rows_per_band` rows of a
// work unit, so everything that depends only on the column -- ((u-cx)*fv)*P00 and ((u-cx)*fv)*P20
// of R/CCalculation.cpp:159-164 -- lives in registers for the whole launch, the per-row
// addressing is one 32-bit add against scalar plane bases, and a wave's loads are 256-byte
// row segments of every plane.  A workgroup covers `bands_per_wg` row bands side by side
// (threads = quads_per_row * bands_per_wg) and strides over (frame-set, band group) units.
// LUT: the first-octant angle table (128.5 KiB) sits in LDS, one workgroup per CU.
// Eligible operands only (slx_strip_eligible): N == 4, dword-aligned planes, W % 4 == 0,
// periods <= 2^14, calibration magnitudes that keep the depth quotient in range.
template <int MODE, int F, bool LUT>
__global__ __launch_bounds__(1024) void slx_strip_kernel(const SlxKParams p)
{
    constexpr bool HAS_GRAY = MODE == SLX_MODE_GRAY_PHASE;
    extern __shared__ __attribute__((aligned(16))) float lds_tab[];
    const unsigned t = threadIdx.x;
    if constexpr (LUT) {
        typedef float vec4 __attribute__((ext_vector_type(4)));
        const vec4 *src = reinterpret_cast<const vec4 *>(p.atan_lut);
        vec4 *dst = reinterpret_cast<vec4 *>(lds_tab);
        for (unsigned i = t; i < SLX_ATAN_LUT_ENTRIES / 4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const unsigned QR = p.quads_per_row;
    const unsigned sub = t / QR;
    const unsigned cq = t - sub * QR;
    const bool lane_ok = sub < p.bands_per_wg;
    const unsigned W = (unsigned)p.width, H = (unsigned)p.height;
    const unsigned row_stride = (unsigned)p.row_stride;

    // column constants (a6): a = (u - cx)*fv ; aC = a*P00 ; aD = a*P20
    double aC[SLX_QUAD], aD[SLX_QUAD];
#pragma unroll
    for (int j = 0; j < SLX_QUAD; j++) {
        const double uc = (double)(int)(cq * SLX_QUAD + j) - p.cx;
        const double a = uc * p.fv;
        aC[j] = a * p.P00;
        aD[j] = a * p.P20;
    }
    float Tf[F];
#pragma unroll
    for (int f = 0; f < F; f++) Tf[f] = (float)p.period[f];

    for (unsigned unit = blockIdx.x; unit < p.total_units; unit += gridDim.x) {
        const unsigned set = unit / p.units_per_set;
        const unsigned ub = unit - set * p.units_per_set;
        const unsigned row0 = (ub * p.bands_per_wg + sub) * p.rows_per_band;
        const size_t pset = (size_t)set * p.phase_set_stride;
        const size_t gset = (size_t)set * p.gray_set_stride;
        double *zset = p.z + (size_t)set * p.out_set_stride;
        unsigned voff = row0 * row_stride + cq * SLX_QUAD;
        unsigned zoff = row0 * W + cq * SLX_QUAD;

        for (unsigned i = 0; i < p.rows_per_band; i++, voff += row_stride, zoff += W) {
            const unsigned row = row0 + i;
            if (!(lane_ok && row < H)) continue;

            float pix[F][SLX_QUAD];
#pragma unroll
            for (int f = 0; f < F; f++) {
                const uint32_t w0 = *reinterpret_cast<const uint32_t *>(p.phase[f * 4 + 0] + pset + voff);
                const uint32_t w1 = *reinterpret_cast<const uint32_t *>(p.phase[f * 4 + 1] + pset + voff);
                const uint32_t w2 = *reinterpret_cast<const uint32_t *>(p.phase[f * 4 + 2] + pset + voff);
                const uint32_t w3 = *reinterpret_cast<const uint32_t *>(p.phase[f * 4 + 3] + pset + voff);
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++)
                    pix[f][j] = wrapped_pix_4step<LUT>(ubyte(w0, j), ubyte(w1, j), ubyte(w2, j), ubyte(w3, j), Tf[f], lds_tab);
            }

            double U[SLX_QUAD];
            if constexpr (HAS_GRAY) {
                unsigned code[SLX_QUAD] = {0u, 0u, 0u, 0u};
                for (int b = p.gray_bits - 1; b >= 0; b--) {       // MSB first: code = 2*code + bit
                    const uint32_t wa = *reinterpret_cast<const uint32_t *>(p.gray[2 * b] + gset + voff);
                    const uint32_t wb = *reinterpret_cast<const uint32_t *>(p.gray[2 * b + 1] + gset + voff);
#pragma unroll
                    for (int j = 0; j < SLX_QUAD; j++)
                        code[j] = code[j] + code[j] + (ibyte(wa, j) > ibyte(wb, j) ? 1u : 0u);
                }
                const double Td = (double)p.period[0], Sd = (double)p.gray_stripe;
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++) {
                    int bin;
                    if (p.std_gray) {                               // inverse reflected Gray code: prefix xor
                        unsigned g = code[j];
                        g ^= g >> 1;
                        g ^= g >> 2;
                        g ^= g >> 4;
                        g ^= g >> 8;
                        bin = (int)g;
                    } else {
                        bin = (int)p.lut[code[j]];
                    }
                    const double grayv = (double)bin * Sd;
                    const double phaseVal = (double)pix[0][j];
                    double ph = phaseVal;
                    if ((bin & 1) == 0) {
                        if (phaseVal > Td * 0.75) ph = phaseVal - Td;
                    } else {
                        if (phaseVal < Td * 0.25) ph = phaseVal + Td;
                        ph = ph - 0.5 * Td;
                    }
                    U[j] = grayv + ph;
                }
            } else {
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++) {
                    double Uf = (double)pix[0][j];
#pragma unroll
                    for (int f = 1; f < F; f++) {
                        int k;
                        Uf = unwrap_stage<true>(Uf, (double)pix[f][j], p.period[f], p.inv_period[f], p.half_biased[f], k);
                    }
                    U[j] = Uf;
                }
            }

            const double vc = (double)((int)row + p.row_offset) - p.cy;
            const double vf = vc * p.fu;
            const double tvC = vf * p.P01, tvD = vf * p.P21;
            typedef double vec2 __attribute__((ext_vector_type(2)));
            double z[SLX_QUAD];
#pragma unroll
            for (int j = 0; j < SLX_QUAD; j++) {
                const double cC = (aC[j] + tvC) + p.K1;
                const double cD = (aD[j] + tvD) + p.K2;
                z[j] = tri_depth<true>(U[j], cC, cD, p.cA, p.cB, p.fov_min, p.fov_max, true);
            }
            vec2 *d = reinterpret_cast<vec2 *>(zset + zoff);
            d[0] = vec2{z[0], z[1]};
            d[1] = vec2{z[2], z[3]};
        }
    }
}

// One workgroup per |max| value, one thread per |min| <= |max|.
__global__ __launch_bounds__(256) void slx_atan_lut_init_kernel(float *table)
{
    const unsigned mx = blockIdx.x, mn = threadIdx.x;
    if (mn > mx) return;
    table[mx * (mx + 1) / 2 + mn] = octant_angle((float)mn, __builtin_fmaxf((float)mx, 1.0f));
}

typedef void (*kernel_fn)(const SlxKParams);

template <int MODE, int F>
kernel_fn pick2(bool n4, bool aux)
{
    if (n4) return aux ? slx_fused_kernel<MODE, F, true, true> : slx_fused_kernel<MODE, F, true, false>;
    return aux ? slx_fused_kernel<MODE, F, false, true> : slx_fused_kernel<MODE, F, false, false>;
}

template <int MODE>
kernel_fn pick_f(int F, bool n4, bool aux)
{
    switch (F) {
    case 1: return pick2<MODE, 1>(n4, aux);
    case 2: return pick2<MODE, 2>(n4, aux);
    case 3: return pick2<MODE, 3>(n4, aux);
    case 4: return pick2<MODE, 4>(n4, aux);
    }
    return nullptr;
}

kernel_fn pick(int mode, int F, bool n4, bool aux)
{
    switch (mode) {
    case SLX_MODE_PHASE_ONLY: return pick2<SLX_MODE_PHASE_ONLY, 1>(n4, false);
    case SLX_MODE_GRAY_ONLY: return slx_fused_kernel<SLX_MODE_GRAY_ONLY, 0, true, false>;
    case SLX_MODE_GRAY_PHASE: return pick2<SLX_MODE_GRAY_PHASE, 1>(n4, aux);
    case SLX_MODE_MULTIFREQ: return pick_f<SLX_MODE_MULTIFREQ>(F, n4, aux);
    case SLX_MODE_MULTIFREQ_GRAYMASK: return pick_f<SLX_MODE_MULTIFREQ_GRAYMASK>(F, n4, aux);
    }
    return nullptr;
}

template <int MODE>
kernel_fn pick_strip(int F, bool lut)
{
    switch (F) {
    case 1: return lut ? slx_strip_kernel<MODE, 1, true> : slx_strip_kernel<MODE, 1, false>;
    case 2: return lut ? slx_strip_kernel<MODE, 2, true> : slx_strip_kernel<MODE, 2, false>;
    case 3: return lut ? slx_strip_kernel<MODE, 3, true> : slx_strip_kernel<MODE, 3, false>;
    case 4: return lut ? slx_strip_kernel<MODE, 4, true> : slx_strip_kernel<MODE, 4, false>;
    }
    return nullptr;
}

}  // namespace

int slx_num_variants(void) { return 4; }

int slx_launch_atan_lut_init(float *table, void *stream)
{
    hipLaunchKernelGGL(slx_atan_lut_init_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, table);
    return (int)hipGetLastError();
}

bool slx_strip_eligible(const SlxKParams &kp, int mode, bool aux)
{
    if (aux || !kp.aligned || kp.n_steps != 4) return false;
    if (mode != SLX_MODE_MULTIFREQ && mode != SLX_MODE_GRAY_PHASE) return false;
    if (kp.quads_per_row == 0 || kp.quads_per_row > 1024) return false;
    for (int f = 0; f < kp.n_freq; f++)
        if (kp.period[f] > (1 << 14)) return false;
    if ((unsigned long long)kp.row_stride * (unsigned)kp.height >= (1ull << 32)) return false;   // 32-bit plane offsets
    if ((unsigned long long)kp.width * (unsigned)kp.height >= (1ull << 29)) return false;          // 32-bit output offsets
    // the depth quotient's operands must stay far inside the double range (tri_depth<LEAN>)
    const double big = 0x1p90;
    for (double v : {kp.cA, kp.cB, kp.K1, kp.K2, kp.P00, kp.P01, kp.P20, kp.P21, kp.fu, kp.fv, kp.cx, kp.cy})
        if (!(__builtin_fabs(v) < big)) return false;
    return true;
}

static int launch_generic(const SlxKParams &kp, int mode, bool aux, int n_sets, void *stream)
{
    kernel_fn fn = pick(mode, kp.n_freq, kp.n_steps == 4, aux);
    if (!fn) return (int)hipErrorInvalidValue;
    const unsigned block = 256;
    unsigned long long threads;
    if (mode == SLX_MODE_MULTIFREQ_GRAYMASK) {
        const unsigned long long waves = ((unsigned long long)kp.n_quads + 61ull) / 62ull;
        threads = waves * 64ull;
    } else {
        threads = kp.n_quads;
    }
    const unsigned grid_x = (unsigned)((threads + block - 1) / block);
    if (grid_x == 0 || n_sets <= 0) return (int)hipErrorInvalidValue;
    // operand shapes were validated by the caller (slx_api.cpp: check_launch_shapes)
    hipLaunchKernelGGL(fn, dim3(grid_x, (unsigned)n_sets, 1), dim3(block, 1, 1), 0, (hipStream_t)stream, kp);
    return (int)hipGetLastError();
}

int slx_launch_fused(const SlxKParams &kp_in, int mode, bool aux, int n_sets, int variant, void *stream)
{
    const bool can_strip = slx_strip_eligible(kp_in, mode, aux);
    if (variant == SLX_VARIANT_GENERIC || !can_strip) {
        if (variant == SLX_VARIANT_STRIP || variant == SLX_VARIANT_STRIP_LUT) return (int)hipErrorInvalidValue;
        return launch_generic(kp_in, mode, aux, n_sets, stream);
    }
    const bool lut = variant == SLX_VARIANT_STRIP_LUT || (variant == SLX_VARIANT_AUTO && kp_in.atan_lut != nullptr && n_sets * kp_in.height >= 2048);
    if (lut && !kp_in.atan_lut) return (int)hipErrorInvalidValue;
    SlxKParams kp = kp_in;
    // geometry: as many row bands side by side as fit in 1024 threads
    const unsigned QR = kp.quads_per_row;
    kp.bands_per_wg = 1024u / QR;
    const unsigned threads = ((QR * kp.bands_per_wg + 63u) / 64u) * 64u;
    const unsigned n_cu = 256, wg_per_cu = lut ? 1u : 2u;
    const unsigned target_wgs = n_cu * wg_per_cu;
    // rows per band: enough units to balance the persistent grid (>= 8 per workgroup when the
    // launch is large), never more than 32 rows so that the tail stays short
    const unsigned long long rows_total = (unsigned long long)kp.height * (unsigned)n_sets;
    unsigned rb = (unsigned)(rows_total / ((unsigned long long)kp.bands_per_wg * target_wgs * 8ull));
    if (rb < 1) rb = 1;
    if (rb > 32) rb = 32;
    kp.rows_per_band = rb;
    kp.units_per_set = ((unsigned)kp.height + kp.bands_per_wg * rb - 1) / (kp.bands_per_wg * rb);
    kp.total_units = kp.units_per_set * (unsigned)n_sets;
    const unsigned grid = kp.total_units < target_wgs ? kp.total_units : target_wgs;
    kernel_fn fn = mode == SLX_MODE_MULTIFREQ ? pick_strip<SLX_MODE_MULTIFREQ>(kp.n_freq, lut)
                                              : pick_strip<SLX_MODE_GRAY_PHASE>(1, lut);
    if (!fn || grid == 0) return (int)hipErrorInvalidValue;
    const size_t lds = lut ? sizeof(float) * SLX_ATAN_LUT_ENTRIES : 0;
    if (lut) {
        hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(fn, dim3(grid, 1, 1), dim3(threads, 1, 1), lds, (hipStream_t)stream, kp);
    return (int)hipGetLastError();
}
