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

#include <algorithm>
#include <type_traits>

#include "slx_device.h"
#include "slx_kernels.h"

#pragma clang fp contract(off)

// Timing diagnostics, never part of the product build (tools/build_exp.sh builds tmp_ab/libslx_exp<N>.so with -DSLX_EXP=N): they leave
// VALU work out -- the results are WRONG -- to show what the strip kernel's memory skeleton alone takes:
//   4: no triangulation   8: no temporal unwrap   16: no angle evaluation
#ifndef SLX_EXP
#define SLX_EXP 0
#endif

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

// Two pixels at a time with explicit 2-vectors, so that every multiply / add / fma of the sequence is a packed
// instruction (v_pk_*_f32 retire two lanes' worth per issue slot); compares and selects stay per component.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// saturate(-a) and saturate(a * s) to [0, 1], both halves in one packed instruction
__device__ __forceinline__ f32x2 pk_neg_sat(f32x2 a)
{
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, -1.0 op_sel_hi:[1,0] clamp" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ f32x2 pk_mul_sat(f32x2 a, f32x2 s)
{
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(s));
    return r;
}
// (byte J of a) - (byte J of b) in one VALU slot (SDWA operand selects); hipcc finds this form
// for only some of the 24 differences of a row step.
// Byte J of a minus byte J of b as a float, times 2^-149: the SDWA byte select zero-extends the byte into the float
// operand, where it is the denormal byte * 2^-149, and the subtraction of two denormals is exact (f32 denormals are on:
// .amdhsa_float_denorm_mode_32 3, hipcc's default for gfx9).  One instruction instead of an integer SDWA subtraction
// plus v_cvt_f32_i32; the caller rescales two such differences with one packed multiply.
template <int J>
__device__ __forceinline__ float byte_diff_denorm(uint32_t a, uint32_t b)
{
    float d;
    if constexpr (J == 0)
        asm("v_sub_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (J == 1)
        asm("v_sub_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (J == 2)
        asm("v_sub_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(d) : "v"(a), "v"(b));
    else
        asm("v_sub_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

template <int J>
__device__ __forceinline__ int byte_diff(uint32_t a, uint32_t b)
{
    int d;
    if constexpr (J == 0)
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (J == 1)
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(d) : "v"(a), "v"(b));
    else if constexpr (J == 2)
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(d) : "v"(a), "v"(b));
    else
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// a1 + a2 for N == 4, all in f32, from s2 = I0 - I2 and c2 = I1 - I3 (twice the reference's
// sinValue / cosValue; integers in [-255, 255] held in floats).
// Identities used (each argued in DESIGN.md, each covered by the exhaustive 511x511 tests):
//  * sinValue/cosValue enter cvFastArctan only through their signs and the ratio min/max, and
//    (a/2)/(b/2) == a/b exactly, so the halving is dropped;
//  * for integer 0 <= a <= b <= 255, b >= 1, RN(a/b) == fma(fma(-b,q0,a), r, q0) with
//    r = v_rcp_f32(b), q0 = a*r (the quotient is never within 1/510 ulp of a rounding tie);
//    ax + (float)DBL_EPSILON == ax for ax >= 0.5, and 0/(0+eps) == 0/1;
//  * RN(x/360) by the same residual correction with r = RN(1/360);
//  * (float)((double)q * (double)T) == q*T in f32 (the double product is exact), and
//    (float)((double)pix + 0.5) == pix + 0.5f (the double sum is exact for pix = 0 or >= 2^-11).
// ---- the phase arithmetic on vectors of pixels ---------------------------------------------------------------------
// f32x2 operands make every multiply / add / fma a packed instruction (two pixels per issue slot).  F32x2x2 is two such
// pairs evaluated in LOCKSTEP: every operation is issued for pair a, then for pair b.  Dependent packed operations need a
// wait state between them -- hipcc pads with s_nop, and an s_nop costs the wave an issue turn like a real instruction
// (122 per row before, 14 after) -- and the other pair's independent operation fills it.
struct F32x2x2 {
    f32x2 a, b;
};
__device__ __forceinline__ F32x2x2 operator+(F32x2x2 x, float y) { return {x.a + y, x.b + y}; }
__device__ __forceinline__ F32x2x2 operator-(F32x2x2 x, float y) { return {x.a - y, x.b - y}; }
__device__ __forceinline__ F32x2x2 operator-(float x, F32x2x2 y) { return {x - y.a, x - y.b}; }
__device__ __forceinline__ F32x2x2 operator-(F32x2x2 x) { return {-x.a, -x.b}; }
__device__ __forceinline__ F32x2x2 operator*(F32x2x2 x, F32x2x2 y) { return {x.a * y.a, x.b * y.b}; }
__device__ __forceinline__ F32x2x2 operator*(F32x2x2 x, float y) { return {x.a * y, x.b * y}; }
__device__ __forceinline__ F32x2x2 operator*(float x, F32x2x2 y) { return {x * y.a, x * y.b}; }

__device__ __forceinline__ f32x2 v_fma(f32x2 x, f32x2 y, f32x2 z) { return __builtin_elementwise_fma(x, y, z); }
__device__ __forceinline__ f32x2 v_fma(float x, f32x2 y, float z) { return __builtin_elementwise_fma(f32x2{x, x}, y, f32x2{z, z}); }
__device__ __forceinline__ f32x2 v_fma(float x, f32x2 y, f32x2 z) { return __builtin_elementwise_fma(f32x2{x, x}, y, z); }
__device__ __forceinline__ f32x2 v_fma(f32x2 x, float y, f32x2 z) { return __builtin_elementwise_fma(x, f32x2{y, y}, z); }
__device__ __forceinline__ f32x2 v_fma(f32x2 x, f32x2 y, float z) { return __builtin_elementwise_fma(x, y, f32x2{z, z}); }
__device__ __forceinline__ f32x2 v_abs(f32x2 x) { return {__builtin_fabsf(x.x), __builtin_fabsf(x.y)}; }
__device__ __forceinline__ f32x2 v_max(f32x2 x, f32x2 y) { return {__builtin_fmaxf(x.x, y.x), __builtin_fmaxf(x.y, y.y)}; }
__device__ __forceinline__ f32x2 v_max3(f32x2 x, f32x2 y, float z) { return {__builtin_fmaxf(__builtin_fmaxf(x.x, y.x), z), __builtin_fmaxf(__builtin_fmaxf(x.y, y.y), z)}; }
__device__ __forceinline__ f32x2 v_min(f32x2 x, f32x2 y) { return {__builtin_fminf(x.x, y.x), __builtin_fminf(x.y, y.y)}; }
__device__ __forceinline__ f32x2 v_rcp(f32x2 x) { return {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ f32x2 v_mul_sat(f32x2 x, float s) { return pk_mul_sat(x, f32x2{s, s}); }
__device__ __forceinline__ f32x2 v_neg_sat(f32x2 x) { return pk_neg_sat(x); }

// ---- selects as arithmetic on 2-cycle instructions --------------------------------------------------------------
// tools/valubench.hip: a wave's plain v_mul_f32 / v_add_f32 (also with |x| or clamp modifiers) issue in 2 cycles, v_fma_f32
// in ~3.5, every packed f32 operation, v_cmp, v_cndmask, v_min / v_max in 4.  A compare + select of a pixel pair costs 16
// cycles; the forms below cost 8 or less.
// sat(x * s), one pixel: v_mul_f32 ... clamp (hipcc has no source form for the clamp bit)
__device__ __forceinline__ float mul_sat1(float x, float s)
{
    float r;
    asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(x), "s"(s));
    return r;
}
// 1.0 where |p| > |q|, else 0.0 (both 0 or separated by far more than 2^-60: every caller's operands are)
__device__ __forceinline__ f32x2 v_gt01_abs(f32x2 p, f32x2 q)
{
    return {mul_sat1(__builtin_fabsf(p.x) - __builtin_fabsf(q.x), 0x1p60f), mul_sat1(__builtin_fabsf(p.y) - __builtin_fabsf(q.y), 0x1p60f)};
}
// fma(k, m, |z|) per pixel: v_fma_f32 with the |.| modifier on the addend (VOP3P has no abs; written as asm because hipcc's
// vectoriser otherwise pairs the two fmas into a v_pk_fma_f32 behind two v_and_b32)
__device__ __forceinline__ float fma_absz1(float k, float m, float z)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, |%3|" : "=v"(r) : "s"(k), "v"(m), "v"(z));
    return r;
}
__device__ __forceinline__ f32x2 v_fma_absz(float k, f32x2 m, f32x2 z) { return {fma_absz1(k, m.x, z.x), fma_absz1(k, m.y, z.y)}; }
// |x| * s per pixel: v_mul_f32 with the |.| modifier
__device__ __forceinline__ float absmul1(float x, float s)
{
    float r;
    asm("v_mul_f32_e64 %0, |%1|, %2" : "=v"(r) : "v"(x), "s"(s));
    return r;
}
__device__ __forceinline__ f32x2 v_absmul(f32x2 x, float s) { return {absmul1(x.x, s), absmul1(x.y, s)}; }
// sat(fma(x, y, z)), both halves in one packed instruction
__device__ __forceinline__ f32x2 pk_fma_sat(f32x2 x, f32x2 y, f32x2 z)
{
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
__device__ __forceinline__ f32x2 v_fma_sat(f32x2 x, float y, float z) { return pk_fma_sat(x, f32x2{y, y}, f32x2{z, z}); }

#define SLX_LOCKSTEP1(NAME) __device__ __forceinline__ F32x2x2 NAME(F32x2x2 x) { return {NAME(x.a), NAME(x.b)}; }
SLX_LOCKSTEP1(v_abs)
SLX_LOCKSTEP1(v_rcp)
SLX_LOCKSTEP1(v_neg_sat)
#undef SLX_LOCKSTEP1
__device__ __forceinline__ F32x2x2 v_fma(F32x2x2 x, F32x2x2 y, F32x2x2 z) { return {v_fma(x.a, y.a, z.a), v_fma(x.b, y.b, z.b)}; }
__device__ __forceinline__ F32x2x2 v_fma(float x, F32x2x2 y, float z) { return {v_fma(x, y.a, z), v_fma(x, y.b, z)}; }
__device__ __forceinline__ F32x2x2 v_fma(float x, F32x2x2 y, F32x2x2 z) { return {v_fma(x, y.a, z.a), v_fma(x, y.b, z.b)}; }
__device__ __forceinline__ F32x2x2 v_fma(F32x2x2 x, float y, F32x2x2 z) { return {v_fma(x.a, y, z.a), v_fma(x.b, y, z.b)}; }
__device__ __forceinline__ F32x2x2 v_fma(F32x2x2 x, F32x2x2 y, float z) { return {v_fma(x.a, y.a, z), v_fma(x.b, y.b, z)}; }
__device__ __forceinline__ F32x2x2 v_max(F32x2x2 x, F32x2x2 y) { return {v_max(x.a, y.a), v_max(x.b, y.b)}; }
__device__ __forceinline__ F32x2x2 v_max3(F32x2x2 x, F32x2x2 y, float z) { return {v_max3(x.a, y.a, z), v_max3(x.b, y.b, z)}; }
__device__ __forceinline__ F32x2x2 v_min(F32x2x2 x, F32x2x2 y) { return {v_min(x.a, y.a), v_min(x.b, y.b)}; }
__device__ __forceinline__ F32x2x2 v_mul_sat(F32x2x2 x, float s) { return {v_mul_sat(x.a, s), v_mul_sat(x.b, s)}; }
__device__ __forceinline__ F32x2x2 v_gt01_abs(F32x2x2 p, F32x2x2 q) { return {v_gt01_abs(p.a, q.a), v_gt01_abs(p.b, q.b)}; }
__device__ __forceinline__ F32x2x2 v_fma_absz(float k, F32x2x2 m, F32x2x2 z) { return {v_fma_absz(k, m.a, z.a), v_fma_absz(k, m.b, z.b)}; }
__device__ __forceinline__ F32x2x2 v_absmul(F32x2x2 x, float s) { return {v_absmul(x.a, s), v_absmul(x.b, s)}; }
__device__ __forceinline__ F32x2x2 v_fma_sat(F32x2x2 x, float y, float z) { return {v_fma_sat(x.a, y, z), v_fma_sat(x.b, y, z)}; }

// From the angle's first-octant value a in [0, 45] to pix: the 90 / 180 / 360 degree fix-ups of cv::fastAtan2, RN(a / 360) * T + 0.5
// and the wrap (R/CDecodePhase.cpp:67-75), without a compare or a select.
//  * A fix-up "m ? K - a : a" (m in {0, 1}, 0 <= a <= K) is |fma(-K, m, a)|: RN(a - K) = -RN(K - a) (rounding is symmetric), and
//    with m = 0 the fma returns a itself.  The |.| rides as a source modifier on the next fix-up's v_fma_f32 and, after the last
//    one, on the multiply by T: everything between -- a * RN(1/360) and its residual correction -- is odd in a, so the sign is
//    carried through and dropped at the end.  mgt: 1 where the sine term is the larger magnitude; mc, ms: 1 where the cosine /
//    sine term is negative.
//  * The wrap "pix > T ? pix - T : pix": m = sat((pix - T) 2^60) in ONE fma (pix 2^60 and T 2^60 are exact, and so is their
//    difference wherever it is positive: pix - T is then a multiple of ulp(pix) in (0, 0.5]); fma(-T, m, pix) is RN(pix - T) or pix.
template <typename V>
__device__ __forceinline__ V pix_from_octant_angle(V a, V mgt, V mc, V ms, float Tf)
{
    const V u = v_fma(-90.f, mgt, a);                                // |u| = 90 - a or a
    const V t = v_fma_absz(-180.f, mc, u);                           // |t| = 180 - |u| or |u|
    const V r = v_fma_absz(-360.f, ms, t);                           // |r| = 360 - |t| or |t|: the angle, sign to be dropped
    const V d0 = r * kInv360;
    const V d = v_fma(v_fma(-360.f, d0, r), kInv360, d0);            // +-RN(angle / 360), see the identities above
    V pix = v_absmul(d, Tf);
    pix = pix + 0.5f;
    const V mw = v_fma_sat(pix, 0x1p60f, -Tf * 0x1p60f);
    return v_fma(-Tf, mw, pix);
}

// a1 for 4 steps from the two differences (sine term s2 = g0 - g2, cosine term c2 = g1 - g3).
// SCALED: s2, c2 carry the integer differences times 2^-23 (byte_diff_denorm below).  Every use of them is invariant
// under a power-of-two scale that stays inside the normal range -- max/min/compare, v_rcp_f32 (a function of the
// mantissa), the quotient and its residual, the sign tests -- so the result is the same bit for bit; only the
// 0/0 guard and the saturation factor of the sign tests move with the scale.
template <bool SCALED, typename V>
__device__ __forceinline__ V wrapped_pix_from_diffs(V s2, V c2, float Tf)
{
#if SLX_EXP & 16
    return v_fma(s2, c2, Tf);              // TIMING DIAGNOSTIC ONLY (wrong results): no angle
#endif
    constexpr float kGuard = SCALED ? 0x1p-23f : 1.0f;
    const V as = v_abs(s2), ac = v_abs(c2);
    const V mx = v_max3(as, ac, kGuard);
    const V mn = v_min(as, ac);
    const V r = v_rcp(mx);
    const V q0 = mn * r;
    const V c = v_fma(v_fma(-mx, q0, mn), r, q0);                   // RN(mn / mx)
    const V cc = c * c;
    const V a = (((kP7 * cc + kP5) * cc + kP3) * cc + kP1) * c;
    const V mc = SCALED ? v_mul_sat(c2, -0x1p60f) : v_neg_sat(c2);
    const V ms = SCALED ? v_mul_sat(s2, -0x1p60f) : v_neg_sat(s2);
    return pix_from_octant_angle(a, v_gt01_abs(s2, c2), mc, ms, Tf);
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
    double zz;
    if constexpr (LEAN) {
        // -(num/den) == num/(-den), and -den = cD U - cC bit for bit when den != 0 (negation commutes with the
        // rounding); this saves the sign flip of the quotient.  The host guarantees |num|, |den| < 2^111
        // (slx_fast_arith_ok) and num is 0 or >= one ulp of cA, so the unscaled sequence can only go wrong when den is
        // zero or denormal -- and then it yields NaN, never a wrong finite value.  NaN -> the literal expression.
        const double nden = cD * Uv - cC;
        zz = div_f64_inrange(num, nden);
        if (__builtin_expect(zz != zz, 0)) zz = -num / (cC - cD * Uv);
    } else {
        zz = -num / (cC - cD * Uv);
    }
    // one select for the three reasons to drop the depth (a NaN depth fails no test and stays, as in the reference)
    const bool drop = (zz < fov_min) | (zz > fov_max) | (Uv == 0.0) | !valid;
    return drop ? 0.0 : zz;
}

// a5 (R/CCalculation.cpp:570-587) for one pixel, cheaply and bit-identically.  The reference:
//   even stripe:  phase > 0.75 T ? phase - T : phase            odd stripe:  (phase < 0.25 T ? phase + T : phase) - 0.5 T;      U = gray + that
// with gray = (double)bin * S and phase the float pix widened to double.  Every one of those double operations is EXACT except the last
// (pix is a multiple of 2^-24 below 2^15, T an integer < 2^22: phase +- T and - T/2 need < 40 bits), so U = RN(gray + phase + c) with
// c in {0, -T, +T/2, -T/2} -- ONE rounding of an exact real.  Here: the two comparisons in f32 (0.25 T and 0.75 T are exact floats, pix IS a
// float), c chosen by 32-bit selects, base = gray + c formed exactly in double (|gray| < 2^48: bin is 16-bit, S 32-bit), U = RN(base + phase):
// the same real, the same rounding.  Saves the two f64 compares and three 64-bit selects per pixel of the literal select chain.
__device__ __forceinline__ double merge_gray_phase(int bin, float pix, float q25, float q75, float Tf, float hTf, double Sd)
{
    const bool odd = (bin & 1) != 0;
    const bool shift = odd ? (pix < q25) : (pix > q75);
    const float c = odd ? (shift ? hTf : -hTf) : (shift ? -Tf : 0.0f);
    const double base = (double)bin * Sd + (double)c;
    return base + (double)pix;
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
                // byte differences as rescaled denormals, the quad's two pixel pairs in lockstep: see wrapped_pix_from_diffs
                const f32x2 kUp = {0x1p126f, 0x1p126f};
                const F32x2x2 px = wrapped_pix_from_diffs<true>(
                    F32x2x2{f32x2{byte_diff_denorm<0>(w0, w2), byte_diff_denorm<1>(w0, w2)} * kUp, f32x2{byte_diff_denorm<2>(w0, w2), byte_diff_denorm<3>(w0, w2)} * kUp},
                    F32x2x2{f32x2{byte_diff_denorm<0>(w1, w3), byte_diff_denorm<1>(w1, w3)} * kUp, f32x2{byte_diff_denorm<2>(w1, w3), byte_diff_denorm<3>(w1, w3)} * kUp}, Tf);
                pix[f][0] = px.a.x;
                pix[f][1] = px.a.y;
                pix[f][2] = px.b.x;
                pix[f][3] = px.b.y;
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
                    int k;
                    if (p.fast_arith) {
                        Uf = unwrap_stage<true>(Uf, pf, p.period[f], p.inv_period[f], p.half_biased[f], k);
                    } else {
                        k = (int)__builtin_floor((Uf - pf) / (double)p.period[f] + 0.5);
                        Uf = pf + (double)(k * p.period[f]);
                    }
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
            const double zz = p.fast_arith ? tri_depth<true>(Uv, cC, cD, p.cA, p.cB, p.fov_min, p.fov_max, valid[j])
                                           : tri_depth<false>(Uv, cC, cD, p.cA, p.cB, p.fov_min, p.fov_max, valid[j]);
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


// R/CDecodePhase.cpp:67-75 for arbitrary float sums (x1, N = 8 in the strip kernel), all in f32.
//  * the angle's division n/d (0 <= n <= d, d in [2^-53, 2^12]) is hipcc's f32 sequence (v_rcp_f32, one Newton step,
//    two residual corrections) without its range scaling and special-case fix-up, which are no-ops in this range;
//  * the casts through double are exact or round identically (the double product of a float and an integer < 2^24
//    is exact; the double sum with 0.5 is either exact or rounds to the float the f32 add gives), and RN(x/360)
//    comes from one residual correction (x/360 is never within 1/90 ulp of a rounding tie);
//  * the saturating tests: sat(x * -2^60) is the sign test (|x|, |y| are 0 or >= 2^-26, see next), and sat((|y| - |x|) 2^60) the
//    "sine term larger" test -- right as long as a NON-ZERO difference of the magnitudes is >= 2^-60.  It is >= 2^-26: every term
//    of the two sums is a multiple of 2^-24 (a byte value, or RN(g r) with r ~ 0.7071, which is 0 or >= 0.7 and so a multiple of
//    its ulp >= 2^-24), f32 sums of multiples of 2^-24 below 2^11 are again multiples of 2^-24 (exact below 1, rounded to a coarser
//    grid above), and the scale wscale = 2/N = 2^-2 makes them multiples of 2^-26.  The bound is on the terms, not on |x|, |y|:
//    the sums may cancel to anything, but never to a non-zero value below the grid.  A change of wscale or of the weights must
//    re-derive it (tests/test_gpu_parity.py::test_eight_step_sine_cosine_near_ties sits on the equal-magnitude directions).
template <typename V>
__device__ __forceinline__ V pix_tail_inrange(V y, V x, float Tf)
{
    const V ax = v_abs(x), ay = v_abs(y);
    const V mx = v_max(ax, ay), mn = v_min(ax, ay);
    const V dd = mx + kEps;
    V r = v_rcp(dd);
    r = v_fma(v_fma(-dd, r, 1.f), r, r);
    V q = mn * r;
    q = v_fma(v_fma(-dd, q, mn), r, q);
    const V c = v_fma(v_fma(-dd, q, mn), r, q);
    const V cc = c * c;
    const V a = (((kP7 * cc + kP5) * cc + kP3) * cc + kP1) * c;
    return pix_from_octant_angle(a, v_gt01_abs(y, x), v_mul_sat(x, -0x1p60f), v_mul_sat(y, -0x1p60f), Tf);
}

// a3 for the four pixels of a quad at once (R/CDecodeGray.cpp:150-176: bit = pattern > inverse, ties -> 0), on the dwords
// of a plane pair, with 2-cycle integer operations instead of a compare + select per pixel:
//   t = (y | 0x80..) - (x & 0x7f..)   per byte 0x80 + ylow - xlow, never a borrow between bytes: bit 7 = (ylow >= xlow)
//   e = x ^ y                         bit 7 set: the top bits differ, and the byte whose top bit is set is the larger
//   bit 7 of (e ? y : t) = (y >= x) = NOT (x > y)
// The caller collects NOT(bit) and inverts once.
__device__ __forceinline__ uint32_t swar_ge_u8_bit7(uint32_t y, uint32_t x)
{
    const uint32_t t = (y | 0x80808080u) - (x & 0x7f7f7f7fu);
    const uint32_t e = x ^ y;
    return (e & y) | (~e & t);                                       // v_bfi_b32
}
// One more bit plane into the running code of four pixels: the new bit enters at bit 7 of each byte and the older ones move
// down, so after G <= 8 planes, taken LSB first, bit b sits at position 8 - G + b and nothing ever crossed a byte boundary.
__device__ __forceinline__ uint32_t swar_push_bit7(uint32_t acc, uint32_t bit7)
{
    return (acc >> 1) | (bit7 & 0x80808080u);                        // v_lshrrev + v_and_or
}
// The four G-bit codes (one per byte) from the collected NOT-bits.
__device__ __forceinline__ uint32_t swar_finish_code(uint32_t acc_not, int G)
{
    return ~(acc_not >> (8 - G)) & (0x01010101u * ((1u << G) - 1u));
}
// Inverse reflected Gray code of four codes of at most 8 bits, one per byte: prefix xor inside every byte.
__device__ __forceinline__ uint32_t swar_gray_to_binary_u8(uint32_t g)
{
    g ^= (g >> 1) & 0x7f7f7f7fu;
    g ^= (g >> 2) & 0x3f3f3f3fu;
    g ^= (g >> 4) & 0x0f0f0f0fu;
    return g;
}

// Wave-uniform values that hipcc's divergence analysis no longer proves uniform once they are carried around a loop and updated
// under a (uniform) condition: told so, they stay in scalar registers -- a buffer descriptor must be in SGPRs.
__device__ __forceinline__ unsigned uniform_u32(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *ptr)
{
    const unsigned long long b = reinterpret_cast<unsigned long long>(ptr);
    return reinterpret_cast<T *>((unsigned long long)uniform_u32((unsigned)b) | ((unsigned long long)uniform_u32((unsigned)(b >> 32)) << 32));
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant expression inside the
// body (the immediate offset of a buffer load must be one)
template <int N, int K = 0, typename Fn>
__device__ __forceinline__ void static_for(Fn &&f)
{
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        static_for<N, K + 1>(f);
    }
}

// ------------------------------------------------------------------------------------------
// Fast path: waves walking column strips, fringe stack staged through LDS.
//
// A work item is 64 quad columns x `rows_per_lane` rows of one frame-set, one item per wave, far
// more waves than the chip holds (the dispatcher back-fills SIMDs as waves retire).  Every lane owns
// one quad column (4 adjacent pixels) and walks down its rows, so what depends only on the column
// -- (u-cx)*fv of R/CCalculation.cpp:159-164 -- stays in registers and the per-row addressing is a
// handful of adds.  When the quads of a row are not a multiple of 64, `interleave` consecutive rows
// are laid end to end (interleave * quads_per_row is) and a lane walks rows with that stride, so no
// lane idles.
//   in : one row of the fringe stack = NP dwords per lane goes HBM -> LDS by DMA (global_load_lds,
//        no VGPRs), two rows ahead of the row being decoded, through a two-slot ring per wave;
//   out: a row's depth is written to LDS as the lanes have it (pixels 4l..4l+3) and read back in
//        store order (lane l next to lane l-1), then stored nontemporally one step late, so that
//        the wait for a row's DMA never waits for a store (tools/membench.hip: lane-contiguous
//        nontemporal stores move this traffic mix at 6.56 TB/s, 32-byte-stride pairs at 5.86).
// hipcc does not track LDS-DMA in its s_waitcnt insertion: the vmcnt waits here are explicit.
// Eligible operands only (slx_strip_eligible): N == 4, dword-aligned planes, W % 4 == 0,
// periods <= 2^14, calibration magnitudes that keep the depth quotient in range.
struct StripPos {
    unsigned set, row, cq;             // frame-set, first row and quad column of this lane
    unsigned out_row[2];               // store slot k of this lane: first row ...
    unsigned out_off[2];               // ... and offset in doubles within the frame-set's depth map
};

// Store slot s = k*64 + lane of store instruction k holds pixels 2s, 2s+1 of the wave's pixel run, i.e. half
// of a source lane's quad.  HALO (Gray-mask mode): a wave owns 62 quads and carries one halo quad on
// either side (lanes 0 and 63), so the 3-tap horizontal AND of x3 never leaves the wave; chunks then
// advance by 62 quads and the run a wave stores is the 62 inner quads.
template <bool HALO>
__device__ __forceinline__ StripPos strip_locate(const SlxKParams &p, unsigned item, unsigned items_per_set, unsigned rows_per_lane, unsigned row0,
                                                 bool &lane_valid)
{
    constexpr unsigned SPAN = HALO ? 62u : 64u;
    StripPos s;
    const unsigned lane = threadIdx.x & 63u;
    s.set = item / items_per_set;
    const unsigned rem = item - s.set * items_per_set;
    const unsigned g = rem / p.chunks_per_group;
    const unsigned c = rem - g * p.chunks_per_group;
    const unsigned row_base = row0 + g * rows_per_lane * p.interleave;
    const unsigned total = p.interleave * p.quads_per_row;           // quads of one row group, rows laid end to end
    const int idx_raw = (int)(c * SPAN + lane) - (HALO ? 1 : 0);
    lane_valid = idx_raw >= 0 && (unsigned)idx_raw < total;
    const unsigned idx = lane_valid ? (unsigned)idx_raw : 0u;
    const unsigned sub = idx / p.quads_per_row;
    s.cq = idx - sub * p.quads_per_row;
    s.row = row_base + sub;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const unsigned slot = (unsigned)k * 64u + lane;
        const unsigned vidx = c * SPAN + (slot >> 1);                // the quad whose pixels this slot stores
        const bool ok = vidx < total && (!HALO || slot < 2u * SPAN);
        const unsigned vsub = vidx / p.quads_per_row;
        const unsigned vcq = vidx - vsub * p.quads_per_row;
        s.out_row[k] = ok ? row_base + vsub : 0xFFFFFFFFu;           // never < H: nothing is stored
        s.out_off[k] = (row_base + vsub) * (unsigned)p.width + vcq * SLX_QUAD + (slot & 1u) * 2u;
    }
    return s;
}

// GB > 0: the 2*GB Gray planes ride the DMA ring behind the phase planes (GB = gray_bits, compile time because
// s_waitcnt takes an immediate); GB == 0: Gray planes, if any, are read with ordinary loads inside the step.
// NS: phase-shift steps, 4 (the reference's, planes through the DMA ring) or another compile-time count (x1:
// the planes are read with ordinary loads, F * NS of them would not fit the ring at a useful occupancy).
// AUX: also the optional planes the reference computes beside z -- x, y (R/CCalculation.cpp:756-771), the projector column U,
// the fringe orders k, the validity mask -- for the whole batch.  The f64 planes take the same LDS transpose as z (a second
// 2-KiB staging area, reused plane after plane: a wave's LDS operations execute in order), k and the mask leave straight from
// the lanes that own the pixels (16 / 4 contiguous bytes per lane).  Planes that were not asked for are "stored" against an
// empty buffer descriptor, so every step issues the same number of memory instructions -- which the counted s_waitcnt
// immediates of the DMA ring rely on.
template <int MODE, int F, int GB, int NS, bool AUX>
__global__ __launch_bounds__(256) void slx_strip_kernel(const SlxKParams p)
{
    constexpr bool MASKED = MODE == SLX_MODE_MULTIFREQ_GRAYMASK;
    constexpr bool HAS_GRAY = MODE == SLX_MODE_GRAY_PHASE || MASKED;
    // The ring moves CHUNKS: with 4 steps a chunk is a whole row of the fringe stack, with 8 steps it is one
    // frequency of a row (8 planes), so that the ring stays 4 KiB per wave and 4 waves per SIMD still fit.
    // Gray planes that ride the ring are a chunk of their own (the second of the row): the ring is then
    // max(4 F, 2 GB) planes wide instead of 4 F + 2 GB, which is what lets 4 waves per SIMD fit beside it.
    constexpr bool GRAY_CHUNK = NS == 4 && GB > 0;
    constexpr int CPR = NS == 4 ? (GRAY_CHUNK ? 2 : 1) : F;   // chunks per row
    constexpr int NPH = NS == 4 ? F * 4 : 8;      // planes in a phase chunk
    constexpr int NGR = 2 * GB;                   // planes in the Gray chunk
    constexpr int NPMAX = NPH > NGR ? NPH : NGR;
    constexpr unsigned ROW_DW = NPMAX * 64;       // one ring slot in LDS, dwords per wave
    constexpr int NK = (F > 1 && MODE != SLX_MODE_GRAY_PHASE) ? F - 1 : 0;   // planes of fringe orders
    constexpr int NZ = 2;                         // depth stores per row
    constexpr int NA = AUX ? 6 + NK + 1 : 0;      // store instructions of the optional planes per row: x, y, U (2 each), k, mask
    // the counted waits below are exact only when every vector-memory operation of a step is one this code issues itself
    constexpr bool EXACT_WAITS = GB > 0 || !HAS_GRAY;
    typedef double vec2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) void lds_void;
    // LDS per wave: [fringe-stack ring: 2 chunks x NP planes x 256 B] [2 KiB depth staging]
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
    const unsigned t = threadIdx.x;
    const unsigned lane = t & 63u;
    // wave-uniform by construction; readfirstlane tells hipcc so (scalar addressing, M0 straight from
    // an SGPR, no waterfall loops around the buffer descriptor)
    const unsigned wave_in_wg = __builtin_amdgcn_readfirstlane(t >> 6);
    uint32_t *ring = lds_raw + wave_in_wg * (2u * ROW_DW + 512u + (AUX ? 512u : 0u));
    vec2 *stage = reinterpret_cast<vec2 *>(ring + 2u * ROW_DW);
    vec2 *stage2 = stage + 128;                   // AUX only
    // XCD-aware item order: the dispatcher deals workgroups round-robin over the 8 XCDs (blocks b and b + 8 share
    // an L2), so consecutive items go to ONE XCD: neighbouring chunks of the Gray-mask mode overlap by a halo quad
    // and start at 248-byte multiples, and their shared 128-byte lines are then fetched from HBM once, not twice.
    // Placement only affects speed; any dispatch order gives the same result.
    // The other modes have no reuse and run in plain order, which is what lets the LAST items be the small ones (below).
    // Tiers of items (SlxKParams): which tier this workgroup belongs to, then its place inside the tier.
    unsigned wg = blockIdx.x;
    unsigned RB = p.tier_rows[0], items_per_set = p.tier_items_per_set[0], region_row0 = 0, tier_items = p.tier_items[0], tier_wgs = p.tier_wgs[0];
    unsigned item_base = 0;
#pragma unroll
    for (int t = 1; t < SLX_MAX_TIERS; t++) {
        if (t < (int)p.n_tiers && blockIdx.x >= p.tier_first_wg[t]) {
            wg = blockIdx.x - p.tier_first_wg[t];
            RB = p.tier_rows[t];
            items_per_set = p.tier_items_per_set[t];
            region_row0 = p.tier_row0[t];
            tier_items = p.tier_items[t];
            tier_wgs = p.tier_wgs[t];
            item_base = p.tier_first_wg[t] * (blockDim.x >> 6);
        }
    }
    if (MASKED && !p.plain_order) {                                 // plain order: A/B measurements only (slx_set_tuning)
        const unsigned nb = tier_wgs, q = nb >> 3, r = nb & 7u, x = wg & 7u, within = wg >> 3;
        wg = x * q + (x < r ? x : r) + within;
    }
    unsigned item = wg * (blockDim.x >> 6) + wave_in_wg;
    if (item >= tier_items) return;
    const unsigned item_id = item_base + item;                      // diagnostics (stamps): unique over the launch
    if (p.stamps && lane == 0 && item_id < p.stamp_items) {   // diagnostics only (slx_debug_stamps)
        p.stamps[4 * item_id + 0] = __builtin_amdgcn_s_memtime();
        p.stamps[4 * item_id + 2] = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned W = (unsigned)p.width, H = (unsigned)p.height;
    const unsigned row_stride = (unsigned)p.row_stride;
    const unsigned step_rows = p.interleave;
    const unsigned last_row = H - 1u;
    float Tf[F];
#pragma unroll
    for (int f = 0; f < F; f++) Tf[f] = (float)p.period[f];
    // A VOP3 instruction takes one scalar operand: fma(d, 1/T, 0.5+eps) with both constants in SGPRs costs
    // a v_mov_b64 per use.  Keeping the addend in a VGPR pair for the whole item removes it.
    double hb[F];
#pragma unroll
    for (int f = 0; f < F; f++) {
        hb[f] = p.half_biased[f];
        asm volatile("" : "+v"(hb[f]));
    }
    // The Gray modes run out of scalar registers (a wave has 102): the per-pixel constants of the triangulation live in
    // vector registers there (those kernels have them to spare below the 128 that 4 waves per SIMD allow).
    double kK1 = p.K1, kK2 = p.K2, kcA = p.cA, kcB = p.cB, kfmin = p.fov_min, kfmax = p.fov_max;
    double kinvT[F];
#pragma unroll
    for (int f = 0; f < F; f++) kinvT[f] = p.inv_period[f];
    if constexpr (MASKED && GB > 0 && F <= 3) {
        asm volatile("" : "+v"(kK1), "+v"(kK2), "+v"(kcA), "+v"(kcB), "+v"(kfmin), "+v"(kfmax));
#pragma unroll
        for (int f = 1; f < F; f++) asm volatile("" : "+v"(kinvT[f]));
    }

    bool lane_valid;
    const StripPos pos = strip_locate<MASKED>(p, item, items_per_set, RB, region_row0, lane_valid);
    const size_t pset = (size_t)pos.set * p.phase_set_stride;
    const size_t gset = (size_t)pos.set * p.gray_set_stride;
    double *zset = p.z + (size_t)pos.set * p.out_set_stride;

    // buffer_load ... lds: one 32-bit lane offset for every plane, the plane's offset in an SGPR, no VALU
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.plane_base + pset), 0, 0xFFFFFFFFu, 0x00020000);
    // lane offsets advance by a constant per row: one add (and a clamp for the DMA) instead of a multiply-add
    const unsigned dma_step = step_rows * row_stride;
    const unsigned dma_last = last_row * row_stride + pos.cq * SLX_QUAD;   // rows past the tile: harmless re-read of the last row
    unsigned dma_off = pos.row * row_stride + pos.cq * SLX_QUAD;           // offset of the row of the next chunk to issue
    // Plane offsets are running scalars: first + k * step, advanced by one s_add per load (the asm keeps hipcc from turning
    // them back into one hoisted register per plane: 24 SGPRs in the Gray modes, which spilled into VGPR lanes and came back
    // through v_readlane every row).
    // The Gray planes that ride the ring have a descriptor of their own (base = Gray plane 0 of this frame-set): the two plane groups
    // may then live anywhere in memory -- separate allocations more than 2 GiB apart included (round 4: bench.py's REF launch fell back
    // to ordinary Gray loads, 200 against 166 us, when the allocator happened to place its two tensors that far apart).
    const __amdgpu_buffer_rsrc_t grsrc =
        GB > 0 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.gray[0] + gset), 0, 0xFFFFFFFFu, 0x00020000) : rsrc;
    auto next_plane = [](unsigned &so, unsigned step) { asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc"); };
    auto issue_chunk = [&](unsigned slot, int cc) {                    // DMA of the item's next chunk (chunk cc of its row) into ring[slot]
        const unsigned voff = dma_off < dma_last ? dma_off : dma_last;
        if (cc == CPR - 1) dma_off += dma_step;
        uint32_t *dst = ring + slot * ROW_DW;
        // every byte is read once: nontemporal loads (+1.6 % on config 4) -- except in the Gray-mask mode, whose halo
        // quads are re-read by the neighbouring wave out of L2 (-10 % with nt there)
        constexpr int POLICY = MASKED ? 0 : 2;
        // The LDS address of a DMA is M0 + the instruction's immediate offset + 4 lane, and the immediate is added to the global
        // address as well: with the immediate stepping through the chunk's planes in LDS (256 k) and 256 taken off the running
        // global offset per plane, M0 is written once per chunk instead of once per load (dma_imm: planes >= 256 bytes apart).
        if (GRAY_CHUNK && cc == 1) {
            unsigned so = 0u;
            if (p.dma_imm) {
                static_for<NGR>([&](auto k) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(grsrc, (lds_void *)dst, 4, voff, so, decltype(k)::value * 256, POLICY);
                    if (decltype(k)::value + 1 < NGR) next_plane(so, p.gray_step - 256u);
                });
            } else {
#pragma unroll
                for (int k = 0; k < NGR; k++) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(grsrc, (lds_void *)(dst + k * 64), 4, voff, so, 0, POLICY);
                    if (k + 1 < NGR) next_plane(so, p.gray_step);
                }
            }
        } else {
            unsigned so = p.phase_first + (NS == 4 ? 0u : (unsigned)cc * NPH * p.phase_step);
            if (p.dma_imm) {
                static_for<NPH>([&](auto k) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)dst, 4, voff, so, decltype(k)::value * 256, POLICY);
                    if (decltype(k)::value + 1 < NPH) next_plane(so, p.phase_step - 256u);
                });
            } else {
#pragma unroll
                for (int k = 0; k < NPH; k++) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(dst + k * 64), 4, voff, so, 0, POLICY);
                    if (k + 1 < NPH) next_plane(so, p.phase_step);
                }
            }
        }
    };
    // Depth stores: buffer stores against a descriptor of this frame-set's depth map -- one 32-bit byte offset per store
    // slot that advances by a constant per row, and the hardware's range check drops the rows past the tile (and the
    // slots that store nothing, whose offset stays out of range) without a compare.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(zset, 0, H * W * 8u, 0x00020000);
    unsigned out_boff[2], out_bstep[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const bool ok = pos.out_row[k] != 0xFFFFFFFFu;
        out_boff[k] = ok ? pos.out_off[k] * 8u : 0xFFFFFFF0u;
        out_bstep[k] = ok ? step_rows * W * 8u : 0u;
    }
    auto flush_row = [&](unsigned) {                                   // depth of the oldest unstored row, in store order
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + k * 64 + lane);
            __builtin_amdgcn_raw_buffer_store_b128(v, zrsrc, out_boff[k], 0, 2 /* nt */);
            out_boff[k] += out_bstep[k];
        }
    };

    // The optional planes: a descriptor per plane (an empty one for a plane that was not asked for), the byte offset of the
    // lane's own quad in the 4-byte and 1-byte planes, the reciprocals of the two divisors of a7's second pass.
    __amdgpu_buffer_rsrc_t xrsrc = zrsrc, yrsrc = zrsrc, Ursrc = zrsrc, mrsrc = zrsrc, krsrc[NK > 0 ? NK : 1];
    unsigned own_px = 0x3FFFFFFCu, own_step = 0u;
    double rfu = 0.0, rfv = 0.0;
    if constexpr (AUX) {
        auto plane = [&](const void *base, size_t planes_per_set, size_t q, unsigned elem) {
            const char *b = base ? static_cast<const char *>(base) + ((size_t)pos.set * planes_per_set + q) * p.out_set_stride * elem : nullptr;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(b ? b : reinterpret_cast<const char *>(zset)), 0, b ? H * W * elem : 0u, 0x00020000);
        };
        xrsrc = plane(p.x, 1, 0, 8);
        yrsrc = plane(p.y, 1, 0, 8);
        Ursrc = plane(p.U, 1, 0, 8);
        mrsrc = plane(p.mask, 1, 0, 1);
#pragma unroll
        for (int f = 0; f < NK; f++) krsrc[f] = plane(p.k, NK, f, 4);
        const bool own = lane_valid && (!MASKED || (lane >= 1u && lane <= 62u));
        own_px = own ? pos.row * W + pos.cq * SLX_QUAD : 0x3FFFFFFCu;   // x 4 bytes still fits 32 bits and stays out of range
        own_step = own ? step_rows * W : 0u;
        rfu = slx_refined_rcp_f64(p.fu);
        rfv = slx_refined_rcp_f64(p.fv);
    }

    // a6, column part: a = (u - cx)*fv ; aC = a*P00 ; aD = a*P20
    double aC[SLX_QUAD], aD[SLX_QUAD];
#pragma unroll
    for (int j = 0; j < SLX_QUAD; j++) {
        const double a = ((double)(int)(pos.cq * SLX_QUAD + j) - p.cx) * p.fv;
        aC[j] = a * p.P00;
        aD[j] = a * p.P20;
    }

    const unsigned total_chunks = RB * CPR;
    issue_chunk(0, 0);
    if (total_chunks > 1) issue_chunk(1, 1 % CPR);

    for (unsigned i = 0; i < RB; i++) {
        const unsigned row = pos.row + i * step_rows;
        float pix[F][SLX_QUAD];
        uint32_t gw[GB > 0 ? 2 * GB : 1];

#pragma unroll
        for (int c = 0; c < CPR; c++) {
            const unsigned g = i * CPR + c;
            const unsigned slot = g & 1u;
            // vmcnt retires in issue order, loads and stores alike: the wait for chunk g names how many operations were issued
            // AFTER its DMA -- the DMA of the chunk(s) behind it, the depth stores (NZ) and the optional planes' stores (NA)
            // that fell in between -- and so leaves exactly those in flight.  A smaller count is always safe (it waits for
            // more); the exact one keeps a row's stores from being waited for in the step that issued them.
            //   one chunk per row:   L(i) | A(i-2) | Z(i-2) L(i+1) A(i-1) | wait
            //   phase + Gray chunk:  L(i,0) | L(i,1) A(i-1) | wait0      L(i,1) | A(i-1) Z(i-1) L(i+1,0) | wait1
            //   8 steps, F chunks:   c = 0: L(i,1) A(i-1)   c = 1: A(i-1) Z(i-1) L(next)   c >= 2: L(next)
            // (the first steps of an item have fewer stores behind them: i selects the variant)
            if (g + 1 >= total_chunks) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (!EXACT_WAITS) {
                if (GRAY_CHUNK && c == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NGR) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH) : "memory");
            } else if (CPR == 1) {
                if (i >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH + NZ + 2 * NA) : "memory");
                else if (i == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH + NA) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH) : "memory");
            } else if (GRAY_CHUNK) {
                if (c == 0) {
                    if (i >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NGR + NA) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NGR) : "memory");
                } else {
                    if (i >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH + NZ + NA) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH) : "memory");
                }
            } else {
                if (c == 0 && i >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH + NA) : "memory");
                else if (c == 1 && i >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH + NZ + NA) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH) : "memory");
            }
            if (c == 0 && i > 0) flush_row(i - 1);                      // last row's stores, one step late
            if (row < H) {
                const uint32_t *src = ring + slot * ROW_DW + lane;
                if constexpr (NS != 4) {                                // x1: weighted sums, k ascending, no contraction
                    static_assert(NS == 8 || NS == 4, "the x1 fast path is written for 8 steps");
                    // 8 steps: weights (cos, sin)(k pi/4) = (1,0) (r,r) (0,1) (-r,r) (-1,0) (-r,-r) (0,-1) (r,-r), r = wy[1]
                    // (checked on the host).  Adding g*0 changes nothing and g*(+-1) is exact, so the k-ascending sums are
                    //   sy = ((((g0 + m1) - m3) - g4) - m5) + m7,   sx = ((((m1 + g2) + m3) - m5) - g6) - m7,   m_k = RN(g_k r)
                    float sy[SLX_QUAD], sx[SLX_QUAD];
                    uint32_t w[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) w[k] = src[k * 64];
                    const float r = p.wy[1];
#pragma unroll
                    for (int j = 0; j < SLX_QUAD; j++) {
                        const float m1 = ubyte(w[1], j) * r, m3 = ubyte(w[3], j) * r, m5 = ubyte(w[5], j) * r, m7 = ubyte(w[7], j) * r;
                        sy[j] = ((((ubyte(w[0], j) + m1) - m3) - ubyte(w[4], j)) - m5) + m7;
                        sx[j] = ((((m1 + ubyte(w[2], j)) + m3) - m5) - ubyte(w[6], j)) - m7;
                    }
                    const F32x2x2 px = pix_tail_inrange(F32x2x2{f32x2{sy[0], sy[1]} * p.wscale, f32x2{sy[2], sy[3]} * p.wscale},
                                                        F32x2x2{f32x2{sx[0], sx[1]} * p.wscale, f32x2{sx[2], sx[3]} * p.wscale}, Tf[c]);
                    pix[c][0] = px.a.x;
                    pix[c][1] = px.a.y;
                    pix[c][2] = px.b.x;
                    pix[c][3] = px.b.y;
                }
#pragma unroll
                for (int f = 0; f < ((NS == 4 && c == 0) ? F : 0); f++) {
                    const uint32_t w0 = src[(f * 4 + 0) * 64], w1 = src[(f * 4 + 1) * 64];
                    const uint32_t w2 = src[(f * 4 + 2) * 64], w3 = src[(f * 4 + 3) * 64];
                    // differences as denormals (x 2^-149), rescaled to x 2^-23 by one packed multiply per pair
                    const f32x2 kUp = {0x1p126f, 0x1p126f};
                    const F32x2x2 px = wrapped_pix_from_diffs<true>(
                        F32x2x2{f32x2{byte_diff_denorm<0>(w0, w2), byte_diff_denorm<1>(w0, w2)} * kUp, f32x2{byte_diff_denorm<2>(w0, w2), byte_diff_denorm<3>(w0, w2)} * kUp},
                        F32x2x2{f32x2{byte_diff_denorm<0>(w1, w3), byte_diff_denorm<1>(w1, w3)} * kUp, f32x2{byte_diff_denorm<2>(w1, w3), byte_diff_denorm<3>(w1, w3)} * kUp}, Tf[f]);
                    pix[f][0] = px.a.x;
                    pix[f][1] = px.a.y;
                    pix[f][2] = px.b.x;
                    pix[f][3] = px.b.y;
                }
                if (GRAY_CHUNK && c == 1) {
#pragma unroll
                    for (int k = 0; k < NGR; k++) gw[k] = src[k * 64];
                }
                // the slot is free once it has been read: chunk g+2 goes into it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (g + 2 < total_chunks) issue_chunk(slot, (c + 2) % CPR); // rows past the tile: keeps the DMA count per step fixed
        }

        double z[SLX_QUAD], U[SLX_QUAD];
        int kf[NK > 0 ? NK : 1][SLX_QUAD];
        if constexpr (MASKED || AUX) {                                  // the mask pass / the optional planes below run for every lane
#pragma unroll
            for (int j = 0; j < SLX_QUAD; j++) z[j] = 0.0, U[j] = 0.0;
#pragma unroll
            for (int f = 0; f < NK; f++)
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++) kf[f][j] = 0;
        }
        int v0[SLX_QUAD] = {1, 1, 1, 1};                                // x3: lanes without pixels never veto
        int okq[SLX_QUAD] = {1, 1, 1, 1};                               // the mask plane: 1 outside the Gray-mask mode
        if (row < H) {
            int bin[SLX_QUAD] = {0, 0, 0, 0};
            if constexpr (HAS_GRAY) {
                const unsigned voff = row * row_stride + pos.cq * SLX_QUAD;
                unsigned code[SLX_QUAD] = {0u, 0u, 0u, 0u};
                const int G = GB > 0 ? GB : p.gray_bits;
                if (GB > 0 || G <= 8) {
                    // a3 + the bit-pack of a4 on the whole quad: the four codes live in the four bytes of one register
                    uint32_t acc = 0u;
                    if constexpr (GB > 0) {
#pragma unroll
                        for (int b = 0; b < GB; b++) acc = swar_push_bit7(acc, swar_ge_u8_bit7(gw[2 * b + 1], gw[2 * b]));
                    } else {
                        for (int b = 0; b < G; b++) {
                            const uint32_t wa = *reinterpret_cast<const uint32_t *>(p.gray[2 * b] + gset + voff);
                            const uint32_t wb = *reinterpret_cast<const uint32_t *>(p.gray[2 * b + 1] + gset + voff);
                            acc = swar_push_bit7(acc, swar_ge_u8_bit7(wb, wa));
                        }
                    }
                    uint32_t code4 = swar_finish_code(acc, G);
                    if (p.std_gray) code4 = swar_gray_to_binary_u8(code4);   // lut[gray] = bin is the reflected code's inverse (one uniform branch)
#pragma unroll
                    for (int j = 0; j < SLX_QUAD; j++) code[j] = (code4 >> (8 * j)) & 0xffu;
                } else {
                    for (int b = G - 1; b >= 0; b--) {                  // more than 8 bits: per pixel, MSB first: code = 2*code + bit
                        const uint32_t wa = *reinterpret_cast<const uint32_t *>(p.gray[2 * b] + gset + voff);
                        const uint32_t wb = *reinterpret_cast<const uint32_t *>(p.gray[2 * b + 1] + gset + voff);
#pragma unroll
                        for (int j = 0; j < SLX_QUAD; j++)
                            code[j] = code[j] + code[j] + (ibyte(wa, j) > ibyte(wb, j) ? 1u : 0u);
                    }
                    if (p.std_gray) {
#pragma unroll
                        for (int j = 0; j < SLX_QUAD; j++) {
                            unsigned g = code[j];
                            g ^= g >> 1;
                            g ^= g >> 2;
                            g ^= g >> 4;
                            g ^= g >> 8;
                            code[j] = g;
                        }
                    }
                }
                if (p.std_gray) {
#pragma unroll
                    for (int j = 0; j < SLX_QUAD; j++) bin[j] = (int)code[j];
                } else {
#pragma unroll
                    for (int j = 0; j < SLX_QUAD; j++) bin[j] = (int)p.lut[code[j]];
                }
            }
            if constexpr (MODE == SLX_MODE_GRAY_PHASE) {
                // a5 (merge_gray_phase: one rounding of the same exact real as the reference's chain of exact double operations)
                const double Sd = (double)p.gray_stripe;
                const float Tm = Tf[0], q25 = 0.25f * Tm, q75 = 0.75f * Tm, hT = 0.5f * Tm;
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++) U[j] = merge_gray_phase(bin[j], pix[0][j], q25, q75, Tm, hT, Sd);
            } else {
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++) {
#if SLX_EXP & 8
                    U[j] = (double)((pix[0][j] + pix[F - 1][j]) + pix[F > 1 ? 1 : 0][j]);   // TIMING DIAGNOSTIC ONLY (wrong results): no unwrap
#else
                    double Uf = (double)pix[0][j];
#pragma unroll
                    for (int f = 1; f < F; f++) {
                        int k;
                        Uf = unwrap_stage<true>(Uf, (double)pix[f][j], p.period[f], kinvT[f], hb[f], k);
                        if constexpr (AUX && NK > 0) kf[f - 1][j] = k;
                    }
                    U[j] = Uf;
#endif
                }
            }
            if constexpr (MASKED) {                                     // x3, stripe agreement of this lane's pixels
                const double Sd = (double)p.gray_stripe;
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++) {
                    const bool ok = __builtin_fabs(U[j] - ((double)bin[j] * Sd + Sd * 0.5)) <= Sd;
                    v0[j] = (!lane_valid || ok) ? 1 : 0;
                }
            }

            {
                const double vc = (double)((int)row + p.row_offset) - p.cy;
                const double vf = vc * p.fu;
                const double tvC = vf * p.P01, tvD = vf * p.P21;
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++) {
                    const double cC = (aC[j] + tvC) + kK1;
                    const double cD = (aD[j] + tvD) + kK2;
#if SLX_EXP & 4
                    z[j] = U[j] + cC;          // TIMING DIAGNOSTIC ONLY (wrong results): no triangulation
#else
                    z[j] = tri_depth<true>(U[j], cC, cD, kcA, kcB, kfmin, kfmax, true);
#endif
                }
            }
            if constexpr (!MASKED) {
                // stage this row's depth; it is stored (slot order) at the top of the next step.  Rows past the
                // tile stage nothing and are never stored.
                stage[2 * lane + 0] = vec2{z[0], z[1]};
                stage[2 * lane + 1] = vec2{z[2], z[3]};
            }
        }
        if constexpr (MASKED) {
            // x3: 3-tap horizontal AND.  Pixel u0-1 lives in lane-1, pixel u0+4 in lane+1 (DPP wave shifts; every
            // lane is active here); lanes 0 and 63 are the halo quads and store nothing.
            const int left = __builtin_amdgcn_update_dpp(1, v0[SLX_QUAD - 1], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const int right = __builtin_amdgcn_update_dpp(1, v0[0], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
            const int u0 = (int)(pos.cq * SLX_QUAD);
#pragma unroll
            for (int j = 0; j < SLX_QUAD; j++) {
                int ok = v0[j];
                const int l = j == 0 ? left : v0[j > 0 ? j - 1 : 0];
                const int r = j == SLX_QUAD - 1 ? right : v0[j < SLX_QUAD - 1 ? j + 1 : 0];
                if (u0 + j > 0) ok &= l;
                if (u0 + j + 1 < (int)W) ok &= r;
                if (!ok) z[j] = 0.0;
                okq[j] = ok;
            }
            if (lane >= 1u && lane <= 62u) {
                stage[2 * (lane - 1u) + 0] = vec2{z[0], z[1]};
                stage[2 * (lane - 1u) + 1] = vec2{z[2], z[3]};
            }
        }
        if constexpr (AUX) {
            // a7, second pass (R/CCalculation.cpp:756-771): x = (z uc)/fu, y = (z vc)/fv from the final depth; then the planes
            // leave in this order, NA store instructions in all, whatever was asked for (see the descriptors above)
            double xo[SLX_QUAD], yo[SLX_QUAD];
            const double vc = (double)((int)row + p.row_offset) - p.cy;
#pragma unroll
            for (int j = 0; j < SLX_QUAD; j++) {
                const double uc = (double)(int)(pos.cq * SLX_QUAD + j) - p.cx;
                xo[j] = slx_div_item_const(z[j] * uc, p.fu, rfu);
                yo[j] = slx_div_item_const(z[j] * vc, p.fv, rfv);
            }
            const unsigned sl = MASKED ? lane - 1u : lane;
            const bool holds = !MASKED || (lane >= 1u && lane <= 62u);
            auto emit_f64 = [&](const double (&v)[SLX_QUAD], __amdgpu_buffer_rsrc_t r) {
                if (holds) {
                    stage2[2 * sl + 0] = vec2{v[0], v[1]};
                    stage2[2 * sl + 1] = vec2{v[2], v[3]};
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const u32x4 t4 = *reinterpret_cast<const u32x4 *>(stage2 + k * 64 + lane);
                    __builtin_amdgcn_raw_buffer_store_b128(t4, r, out_boff[k], 0, 2 /* nt */);   // out_boff: this row's slots (z follows next step)
                }
                __builtin_amdgcn_wave_barrier();
            };
            emit_f64(xo, xrsrc);
            emit_f64(yo, yrsrc);
            emit_f64(U, Ursrc);
#pragma unroll
            for (int f = 0; f < NK; f++) {
                const u32x4 k4 = {(unsigned)kf[f][0], (unsigned)kf[f][1], (unsigned)kf[f][2], (unsigned)kf[f][3]};
                __builtin_amdgcn_raw_buffer_store_b128(k4, krsrc[f], own_px * 4u, 0, 2);
            }
            const unsigned m4 = (unsigned)(okq[0] != 0) | (unsigned)(okq[1] != 0) << 8 | (unsigned)(okq[2] != 0) << 16 | (unsigned)(okq[3] != 0) << 24;
            __builtin_amdgcn_raw_buffer_store_b32(m4, mrsrc, own_px, 0, 2);
            own_px += own_step;
        }
        __builtin_amdgcn_wave_barrier();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    flush_row(RB - 1);
    if (p.stamps && lane == 0 && item_id < p.stamp_items) {
        p.stamps[4 * item_id + 1] = __builtin_amdgcn_s_memtime();
        p.stamps[4 * item_id + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// ------------------------------------------------------------------------------------------
// Stream kernel (round 4): the 3 x 4-step class (Gray-free, depth only) with work handed out in ADDRESS ORDER to resident waves.
//
// What round 4 measured: the strip kernel's launch is its memory skeleton plus a tenth of its VALU time,
// and the skeleton is 13 % faster with 2-row items than with 16-row items -- the memory system likes the chip's accesses to
// advance through ONE frame-set in order -- while the arithmetic needs long items, because what an item sets up (lane geometry
// by integer division, the column terms of cC / cD, descriptors) costs a third of a row.  Here a wave sets that state up once per
// LAUNCH and then takes short items (R rows of one 64-quad chunk column) from a queue until the queue is empty:
//   * queue q = (chunk column c, residue j): its k-th item is row group G = k m + j of the batch's frame-sets laid end to end
//     (G -> frame-set, group by one scalar multiply-high), always of column c -- so the wave's per-column state never changes;
//   * a ticket is ONE scalar atomic (s_atomic_add ... glc: old value to an SGPR, counted by lgkmcnt, no vector register, not in
//     the vmcnt sequence the DMA ring counts) issued just before the step's wait for its DMA chunk, so its latency hides there;
//   * the queues' fronts advance together: at any moment the resident waves hold a band of consecutive row groups of one or two
//     frame-sets, whatever the speed of the single waves (oldest-first arbitration makes them unequal: a slow wave simply takes
//     fewer items) -- the footprint of short items with the start-up cost of one item per wave;
//   * the DMA ring (two chunks ahead), the counted waits, the staged lane-contiguous stores are the strip kernel's, carried
//     across item boundaries: the request side runs two rows ahead of the compute side and crosses into the next item first.
// Counters: one 32-bit word per queue, 128 bytes apart (p.sq_counters); a launch draws exactly K_q + W_q tickets from counter q (K_q
// items, one failing ticket per wave), and the wave that draws the last one puts the counter back to zero (round 6; rounds 4-5 let
// the counters run on and told the kernel the launch's number): every launch starts from zero, the host zeroes the words once when
// it allocates them and after a failed launch, and a captured launch can be replayed.
template <int F>
__global__ __launch_bounds__(256) void slx_stream_kernel(const SlxKParams p)
{
    constexpr int NPH = F * 4;                     // planes of a row = one ring chunk
    constexpr unsigned ROW_DW = NPH * 64;
    constexpr int NZ = 2;
    typedef double vec2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
    const unsigned t = threadIdx.x;
    const unsigned lane = t & 63u;
    const unsigned wave_in_wg = __builtin_amdgcn_readfirstlane(t >> 6);
    uint32_t *ring = lds_raw + wave_in_wg * (2u * ROW_DW + 512u);
    vec2 *stage = reinterpret_cast<vec2 *>(ring + 2u * ROW_DW);

    // ---- this wave's queue
    const unsigned waves_per_wg = blockDim.x >> 6;
    const unsigned wave_id = blockIdx.x * waves_per_wg + wave_in_wg, total_waves = gridDim.x * waves_per_wg;
    const unsigned NQ = p.sq_queues, m = p.sq_m, cpg = p.chunks_per_group;
    const unsigned q = wave_id % NQ;
    const unsigned c = q % cpg, j = q / cpg;
    const unsigned Kq = p.sq_groups_total > j ? (p.sq_groups_total - j + m - 1u) / m : 0u;     // items of this queue
    const unsigned Wq = (total_waves - q + NQ - 1u) / NQ;                                       // waves that poll it
    const unsigned last_ticket = Kq + Wq - 1u;                                                  // a launch draws exactly K_q + W_q tickets from queue q
    unsigned *ctr = p.sq_counters + (size_t)q * 32u;
    auto fetch_issue = [&](unsigned &raw) {        // the ticket arrives with the next s_waitcnt lgkmcnt(0)
        asm volatile("s_mov_b32 %0, 1\n\ts_atomic_add %0, %1, 0x0 glc" : "=&s"(raw) : "s"(ctr) : "memory");
    };
    auto fetch_wait = [&](unsigned &raw) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(raw)::"memory"); };
    // The wave that draws a queue's LAST ticket of the launch (every wave draws until its first failing one: K_q + W_q draws in all, and
    // nobody draws after the last) puts the counter back to zero -- a scalar atomic without return, outside the vmcnt sequence.  Every
    // launch therefore finds its counters at zero, whatever ran before it: nothing to reset from the host between launches or
    // geometries, and a launch captured into a hipGraph can be replayed (rounds 4-5 kept an epoch on the host, which a replay repeats).
    auto reset_behind_last = [&](unsigned raw) {
        if (raw == last_ticket) asm volatile("s_atomic_and %0, %1, 0x0" ::"s"(0u), "s"(ctr) : "memory");
    };

    // ---- per-column state, once per launch
    const unsigned W = (unsigned)p.width, H = (unsigned)p.height;
    const unsigned row_stride = (unsigned)p.row_stride, il = p.interleave, R = p.sq_rows;
    const unsigned QR = p.quads_per_row;
    const unsigned idx = c * 64u + lane;                           // < il * QR: chunks_per_group * 64 is exactly that
    const unsigned sub = idx / QR, cq = idx - sub * QR;            // row within the row group, quad column
    unsigned out_lane[2];                                          // byte offset of store slot k within the group's first row ... (+ sub rows)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const unsigned slot = (unsigned)k * 64u + lane, vidx = c * 64u + (slot >> 1);
        const unsigned vsub = vidx / QR, vcq = vidx - vsub * QR;
        out_lane[k] = (vsub * W + vcq * SLX_QUAD + (slot & 1u) * 2u) * 8u;
    }
    const unsigned dma_lane = sub * row_stride + cq * SLX_QUAD;
    const unsigned dma_last = (H - 1u) * row_stride + cq * SLX_QUAD;     // rows past the tile: harmless re-read of the last row
    const unsigned dma_step = il * row_stride, out_step = il * W * 8u;
    float Tf[F];
#pragma unroll
    for (int f = 0; f < F; f++) Tf[f] = (float)p.period[f];
    double hb[F];
#pragma unroll
    for (int f = 0; f < F; f++) {
        hb[f] = p.half_biased[f];
        asm volatile("" : "+v"(hb[f]));
    }
    double aC[SLX_QUAD], aD[SLX_QUAD];
#pragma unroll
    for (int jx = 0; jx < SLX_QUAD; jx++) {
        const double a = ((double)(int)(cq * SLX_QUAD + jx) - p.cx) * p.fv;
        aC[jx] = a * p.P00;
        aD[jx] = a * p.P20;
    }

    // ---- items: ticket -> (frame-set, first row of the row group)
    struct Item { unsigned valid, set, row_base; };
    auto decode = [&](unsigned raw) {
        Item it;
        const unsigned k = raw;
        it.valid = k < Kq ? 1u : 0u;
        const unsigned G = it.valid ? k * m + j : 0u;
        it.set = p.sq_groups_per_set == 1u ? G : __umulhi(G, p.sq_magic);   // G / groups_per_set (exact for G * groups_per_set < 2^32: slx_plan.cpp)
        it.row_base = (G - it.set * p.sq_groups_per_set) * R * il;
        return it;
    };
    unsigned raw;
    fetch_issue(raw);
    fetch_wait(raw);
    reset_behind_last(raw);
    Item A = decode(raw), B{0u, 0u, 0u};
    if (!A.valid) return;

    // request side (two chunks ahead): its item, rows of it already requested, descriptor, lane offset of the next row to request
    Item D = A;
    unsigned dri = 0;
    __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.plane_base + (size_t)D.set * p.phase_set_stride), 0, 0xFFFFFFFFu, 0x00020000);
    unsigned dma_off = dma_lane + D.row_base * row_stride;
    auto next_plane = [](unsigned &so, unsigned step) { asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc"); };
    auto issue_chunk = [&](unsigned slot) {
        const unsigned voff = dma_off < dma_last ? dma_off : dma_last;
        uint32_t *dst = ring + slot * ROW_DW;
        unsigned so = p.phase_first;
        if (p.dma_imm) {
            static_for<NPH>([&](auto k) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(drsrc, (lds_void *)dst, 4, voff, so, decltype(k)::value * 256, 2 /* nt */);
                if (decltype(k)::value + 1 < NPH) next_plane(so, p.phase_step - 256u);
            });
        } else {
#pragma unroll
            for (int k = 0; k < NPH; k++) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(drsrc, (lds_void *)(dst + k * 64), 4, voff, so, 0, 2 /* nt */);
                if (k + 1 < NPH) next_plane(so, p.phase_step);
            }
        }
        dma_off += dma_step;
        dri = uniform_u32(dri + 1u);
    };
    // the request side steps into item `to` (its first row comes next)
    auto request_moves_to = [&](const Item &to) {
        D = to;
        dri = 0;
        if (to.valid) {
            drsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(uniform_ptr(p.plane_base + (size_t)to.set * p.phase_set_stride)), 0, 0xFFFFFFFFu, 0x00020000);
            dma_off = dma_lane + to.row_base * row_stride;
        }
    };

    // compute side: item A, row ri of it; where its rows are stored
    unsigned ri = 0;
    unsigned row = A.row_base + sub;
    __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(p.z + (size_t)A.set * p.out_set_stride, 0, H * W * 8u, 0x00020000);
    unsigned out_boff[2] = {out_lane[0] + A.row_base * W * 8u, out_lane[1] + A.row_base * W * 8u};
    // the staged row waiting for its (one step late) stores
    __amdgpu_buffer_rsrc_t prsrc = zrsrc;
    unsigned pend_off[2] = {0xFFFFFFF0u, 0xFFFFFFF0u};
    issue_chunk(0);                                                // rows 0 and 1 of the first item (R >= 2)
    issue_chunk(1);
    unsigned ahead = 2;                                            // chunks requested and not yet waited for
    for (unsigned s = 0;; s++) {
        const unsigned slot = s & 1u;
        // Priority while a wave works towards its next DMA request (wait, flush, LDS reads, phase arithmetic, request), none in the
        // f64 tail behind it: the SIMD's arbiter then prefers the waves whose next action puts bytes in flight.  Same-box A/B:
        // C4 x 32 268.8 vs 274.9 us (-2.2 %), C2 x 80 270.9 vs 275.1 (-1.5 %); priority in the tail instead: +1.7 %; priority only up
        // to an EARLIER request (all dwords first, then the request, then the arithmetic): +1 %.
        __builtin_amdgcn_s_setprio(1);
        // The ticket of the item after this one; lands below.  A variable of this step alone: its live range is the window between its
        // issue and its wait and nothing else (a variable carried around the loop is live everywhere, and the allocator then saves and
        // restores its register around whatever needs scalar registers -- harmless after the wait, fatal inside the window;
        // tests/test_kernel_resources.py pins the compiled code)
        unsigned next_raw;
        if (ri == 0) fetch_issue(next_raw);
        // counted wait for chunk s (vmcnt retires in issue order): L(s) | Z(s-2) L(s+1) | wait -- as in slx_strip_kernel
        if (ahead < 2u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (s >= 2u) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH + NZ) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH) : "memory");
        ahead--;
        if (s > 0) {                                               // last row's stores, one step late
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + k * 64 + lane);
                __builtin_amdgcn_raw_buffer_store_b128(v, prsrc, pend_off[k], 0, 2 /* nt */);
            }
        }
        float pix[F][SLX_QUAD];
        if (row < H) {
            const uint32_t *src = ring + slot * ROW_DW + lane;
#pragma unroll
            for (int f = 0; f < F; f++) {
                const uint32_t w0 = src[(f * 4 + 0) * 64], w1 = src[(f * 4 + 1) * 64];
                const uint32_t w2 = src[(f * 4 + 2) * 64], w3 = src[(f * 4 + 3) * 64];
                const f32x2 kUp = {0x1p126f, 0x1p126f};
                const F32x2x2 px = wrapped_pix_from_diffs<true>(
                    F32x2x2{f32x2{byte_diff_denorm<0>(w0, w2), byte_diff_denorm<1>(w0, w2)} * kUp, f32x2{byte_diff_denorm<2>(w0, w2), byte_diff_denorm<3>(w0, w2)} * kUp},
                    F32x2x2{f32x2{byte_diff_denorm<0>(w1, w3), byte_diff_denorm<1>(w1, w3)} * kUp, f32x2{byte_diff_denorm<2>(w1, w3), byte_diff_denorm<3>(w1, w3)} * kUp}, Tf[f]);
                pix[f][0] = px.a.x;
                pix[f][1] = px.a.y;
                pix[f][2] = px.b.x;
                pix[f][3] = px.b.y;
            }
        }
        // the slot is free once it has been read -- and the ticket requested above has arrived with the same wait
        if (ri == 0) {
            fetch_wait(next_raw);
            reset_behind_last(next_raw);
            B = decode(next_raw);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // chunk s + 2: the request side's next row, in this item or the first row of the next one
        if (D.valid && dri == R) request_moves_to(B);
        if (D.valid) {
            issue_chunk(slot);
            ahead++;
        }
        __builtin_amdgcn_s_setprio(0);
        if (row < H) {
            double z[SLX_QUAD];
            const double vc = (double)((int)row + p.row_offset) - p.cy;
            const double vf = vc * p.fu;
            const double tvC = vf * p.P01, tvD = vf * p.P21;
#pragma unroll
            for (int jx = 0; jx < SLX_QUAD; jx++) {
#if SLX_EXP & 8
                double Uf = (double)((pix[0][jx] + pix[F - 1][jx]) + pix[F > 1 ? 1 : 0][jx]);   // TIMING DIAGNOSTIC ONLY: no unwrap
#else
                double Uf = (double)pix[0][jx];
#pragma unroll
                for (int f = 1; f < F; f++) {
                    int kk;
                    Uf = unwrap_stage<true>(Uf, (double)pix[f][jx], p.period[f], p.inv_period[f], hb[f], kk);
                }
#endif
                const double cC = (aC[jx] + tvC) + p.K1;
                const double cD = (aD[jx] + tvD) + p.K2;
#if SLX_EXP & 4
                z[jx] = Uf + cC + cD;                              // TIMING DIAGNOSTIC ONLY: no triangulation
#else
                z[jx] = tri_depth<true>(Uf, cC, cD, p.cA, p.cB, p.fov_min, p.fov_max, true);
#endif
            }
            stage[2 * lane + 0] = vec2{z[0], z[1]};
            stage[2 * lane + 1] = vec2{z[2], z[3]};

        }
        // this row's stores go out at the top of the next step
        prsrc = zrsrc;
        pend_off[0] = out_boff[0];
        pend_off[1] = out_boff[1];
        __builtin_amdgcn_wave_barrier();
        // next row of the item, or the next item
        ri = uniform_u32(ri + 1u);
        if (ri == R) {
            if (!B.valid) break;
            A = B;
            ri = 0;
            row = A.row_base + sub;
            zrsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.z + (size_t)A.set * p.out_set_stride), 0, H * W * 8u, 0x00020000);
            out_boff[0] = out_lane[0] + A.row_base * W * 8u;
            out_boff[1] = out_lane[1] + A.row_base * W * 8u;
        } else {
            row += il;
            out_boff[0] += out_step;
            out_boff[1] += out_step;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + k * 64 + lane);
        __builtin_amdgcn_raw_buffer_store_b128(v, prsrc, pend_off[k], 0, 2 /* nt */);
    }
}

// ------------------------------------------------------------------------------------------
// Stream kernel of the reference's own mode (round 6): SLX_MODE_GRAY_PHASE -- a1 + a3 + a4 + a5 + a7, R/CCalculation.cpp:525-592 + :666-708,
// 6 Gray bits whose 12 planes ride the DMA ring.  The strip kernel serves that mode with 3-row items -- the shortest-lived work items
// of the library, so what an item sets up (lane geometry by integer division, the buffer descriptors, the column terms of cC / cD)
// is paid every third row.  Here, as in slx_stream_kernel, a resident wave sets that up once per LAUNCH and takes R-row items of ONE
// chunk column from its queue (same queues, tickets and counters).  What differs from that kernel:
//   * a row is TWO ring chunks -- its 4 phase planes (slot 0) and its 12 Gray planes (slot 1) -- so the request side runs one row
//     ahead: the phase chunk of row r + 1 is requested once the phase dwords of row r have been read, the Gray chunk likewise;
//   * counted waits (vmcnt retires in issue order):  P(r) G(r) | Z(r-1) P(r+1) | G(r+1) Z(r) ...: the wait for P(r) leaves G(r) in
//     flight (NGR), the wait for G(r) leaves Z(r-1) and P(r+1) (NZ + NPH; the launch's first row has no Z in front of it);
//   * items of ONE row (slx_plan.cpp): REF x 32 164.3 us with 1-row items, 171.3 with 2, 177.6 with 3, 192 with 8 -- the waves of a
//     SIMD wait for their DMA here, and the more often they draw tickets the less they move in step.
// Same-box A/B against the strip kernel (profiles/r06_gstream_ab.log): REF x 32 169.5 -> 164.3 us (-3.1 %), x 16 -3.5 %, x 8 -2.4 %, x 4
// +0.7 %; 55.7 M vector instructions per launch for 59.9 M, the same bytes.  The Gray-MASK mode (x3: 62-quad spans with halo lanes)
// was built on this kernel too, in five versions -- chip-wide queues (1 282 MB read for 885 algorithmic: the two waves that share a
// halo line almost never share an L2), XCD-local queues (1 017 MB), one ticket per workgroup and item handed over through LDS (942 MB)
// -- and none beat the strip kernel's statically aligned items by 2 % on config 3 (best: -1.2 %): not in the tree
// (profiles/experiments/r06_gstream_kernel_gray_mask_mode.patch).
__global__ __launch_bounds__(256) void slx_gstream_kernel(const SlxKParams p)
{
    constexpr int F = 1, GB = 6;
    constexpr int NPH = F * 4, NGR = 2 * GB;
    constexpr int NPMAX = NPH > NGR ? NPH : NGR;
    constexpr unsigned ROW_DW = NPMAX * 64;
    constexpr int NZ = 2;
    typedef double vec2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
    const unsigned t = threadIdx.x;
    const unsigned lane = t & 63u;
    const unsigned wave_in_wg = __builtin_amdgcn_readfirstlane(t >> 6);
    uint32_t *ring = lds_raw + wave_in_wg * (2u * ROW_DW + 512u);
    vec2 *stage = reinterpret_cast<vec2 *>(ring + 2u * ROW_DW);

    // ---- this wave's queue (slx_stream_kernel)
    const unsigned waves_per_wg = blockDim.x >> 6;
    const unsigned wave_id = blockIdx.x * waves_per_wg + wave_in_wg, total_waves = gridDim.x * waves_per_wg;
    const unsigned NQ = p.sq_queues, m = p.sq_m, cpg = p.chunks_per_group;
    const unsigned q = wave_id % NQ;
    const unsigned c = q % cpg, j = q / cpg;
    const unsigned Kq = p.sq_groups_total > j ? (p.sq_groups_total - j + m - 1u) / m : 0u;
    const unsigned Wq = (total_waves - q + NQ - 1u) / NQ;
    const unsigned last_ticket = Kq + Wq - 1u;
    unsigned *ctr = p.sq_counters + (size_t)q * 32u;
    auto fetch_issue = [&](unsigned &raw) {
        asm volatile("s_mov_b32 %0, 1\n\ts_atomic_add %0, %1, 0x0 glc" : "=&s"(raw) : "s"(ctr) : "memory");
    };
    auto fetch_wait = [&](unsigned &raw) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(raw)::"memory"); };
    auto reset_behind_last = [&](unsigned raw) {   // the queue's last ticket of the launch: the counter goes back to zero (slx_stream_kernel)
        if (raw == last_ticket) asm volatile("s_atomic_and %0, %1, 0x0" ::"s"(0u), "s"(ctr) : "memory");
    };

    // ---- per-column state, once per launch
    const unsigned W = (unsigned)p.width, H = (unsigned)p.height;
    const unsigned row_stride = (unsigned)p.row_stride, il = p.interleave, R = p.sq_rows;
    const unsigned QR = p.quads_per_row;
    const unsigned idx = c * 64u + lane;                           // < il * QR: chunks_per_group * 64 is exactly that
    const unsigned sub = idx / QR, cq = idx - sub * QR;            // row within the row group, quad column
    unsigned out_lane[2];                                          // byte offset of store slot k within the group's first row
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const unsigned slot = (unsigned)k * 64u + lane, vidx = c * 64u + (slot >> 1);
        const unsigned vsub = vidx / QR, vcq = vidx - vsub * QR;
        out_lane[k] = (vsub * W + vcq * SLX_QUAD + (slot & 1u) * 2u) * 8u;
    }
    const unsigned dma_lane = sub * row_stride + cq * SLX_QUAD;
    const unsigned dma_last = (H - 1u) * row_stride + cq * SLX_QUAD;     // rows past the tile: harmless re-read of the last row
    const unsigned dma_step = il * row_stride, out_step = il * W * 8u;
    const float Tf = (float)p.period[0];
    // as in slx_strip_kernel: this kernel is short of scalar registers, the per-pixel constants of the triangulation live in vector ones
    double kK1 = p.K1, kK2 = p.K2, kcA = p.cA, kcB = p.cB, kfmin = p.fov_min, kfmax = p.fov_max;
    asm volatile("" : "+v"(kK1), "+v"(kK2), "+v"(kcA), "+v"(kcB), "+v"(kfmin), "+v"(kfmax));
    double aC[SLX_QUAD], aD[SLX_QUAD];
#pragma unroll
    for (int jx = 0; jx < SLX_QUAD; jx++) {
        const double a = ((double)(int)(cq * SLX_QUAD + jx) - p.cx) * p.fv;
        aC[jx] = a * p.P00;
        aD[jx] = a * p.P20;
    }

    // ---- items: ticket -> (frame-set, first row of the row group)
    struct Item { unsigned valid, set, row_base; };
    auto decode = [&](unsigned raw) {
        Item it;
        const unsigned k = raw;
        it.valid = k < Kq ? 1u : 0u;
        const unsigned G = it.valid ? k * m + j : 0u;
        it.set = p.sq_groups_per_set == 1u ? G : __umulhi(G, p.sq_magic);
        it.row_base = (G - it.set * p.sq_groups_per_set) * R * il;
        return it;
    };
    unsigned raw;
    fetch_issue(raw);
    fetch_wait(raw);
    reset_behind_last(raw);
    Item A = decode(raw), B{0u, 0u, 0u};
    if (!A.valid) return;

    // request side (one row = two chunks ahead): its item, rows of it already requested, the two descriptors, lane offset of the next row
    Item D = A;
    unsigned dri = 0;
    __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.plane_base + (size_t)D.set * p.phase_set_stride), 0, 0xFFFFFFFFu, 0x00020000);
    __amdgpu_buffer_rsrc_t dgrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.gray[0] + (size_t)D.set * p.gray_set_stride), 0, 0xFFFFFFFFu, 0x00020000);
    unsigned dma_off = dma_lane + D.row_base * row_stride;
    auto next_plane = [](unsigned &so, unsigned step) { asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc"); };
    auto issue_phase = [&]() {                                     // the request row's phase planes into slot 0 (every byte is read once: nontemporal)
        const unsigned voff = dma_off < dma_last ? dma_off : dma_last;
        uint32_t *dst = ring;
        unsigned so = p.phase_first;
        if (p.dma_imm) {
            static_for<NPH>([&](auto k) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(drsrc, (lds_void *)dst, 4, voff, so, decltype(k)::value * 256, 2 /* nt */);
                if (decltype(k)::value + 1 < NPH) next_plane(so, p.phase_step - 256u);
            });
        } else {
#pragma unroll
            for (int k = 0; k < NPH; k++) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(drsrc, (lds_void *)(dst + k * 64), 4, voff, so, 0, 2 /* nt */);
                if (k + 1 < NPH) next_plane(so, p.phase_step);
            }
        }
    };
    auto issue_gray = [&]() {                                      // ... its Gray planes into slot 1; the request side moves on a row
        const unsigned voff = dma_off < dma_last ? dma_off : dma_last;
        uint32_t *dst = ring + ROW_DW;
        unsigned so = 0u;
        if (p.dma_imm) {
            static_for<NGR>([&](auto k) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(dgrsrc, (lds_void *)dst, 4, voff, so, decltype(k)::value * 256, 2 /* nt */);
                if (decltype(k)::value + 1 < NGR) next_plane(so, p.gray_step - 256u);
            });
        } else {
#pragma unroll
            for (int k = 0; k < NGR; k++) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(dgrsrc, (lds_void *)(dst + k * 64), 4, voff, so, 0, 2 /* nt */);
                if (k + 1 < NGR) next_plane(so, p.gray_step);
            }
        }
        dma_off += dma_step;
        dri = uniform_u32(dri + 1u);
    };
    auto request_moves_to = [&](const Item &to) {
        D = to;
        dri = 0;
        if (to.valid) {
            drsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(uniform_ptr(p.plane_base + (size_t)to.set * p.phase_set_stride)), 0, 0xFFFFFFFFu, 0x00020000);
            dgrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(uniform_ptr(p.gray[0] + (size_t)to.set * p.gray_set_stride)), 0, 0xFFFFFFFFu, 0x00020000);
            dma_off = dma_lane + to.row_base * row_stride;
        }
    };

    // compute side: item A, row ri of it; where its rows are stored
    unsigned ri = 0;
    unsigned row = A.row_base + sub;
    __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(p.z + (size_t)A.set * p.out_set_stride, 0, H * W * 8u, 0x00020000);
    unsigned out_boff[2] = {out_lane[0] + A.row_base * W * 8u, out_lane[1] + A.row_base * W * 8u};
    __amdgpu_buffer_rsrc_t prsrc = zrsrc;
    unsigned pend_off[2] = {0xFFFFFFF0u, 0xFFFFFFF0u};
    issue_phase();                                                 // row 0 of the first item
    issue_gray();
    for (unsigned s = 0;; s++) {
        __builtin_amdgcn_s_setprio(1);                             // (slx_stream_kernel: priority up to the step's last DMA request)
        unsigned next_raw;                                         // the ticket of the item after this one: a variable of this step alone (see slx_stream_kernel)
        if (ri == 0) fetch_issue(next_raw);
        // ---- chunk 0: the row's phase planes.  Behind their DMA only the row's Gray DMA has been issued
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NGR) : "memory");
        if (s > 0) {                                               // last row's stores, one step late
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + k * 64 + lane);
                __builtin_amdgcn_raw_buffer_store_b128(v, prsrc, pend_off[k], 0, 2 /* nt */);
            }
        }
        // No divergent branch between the ticket's issue and its wait: rows past the tile run the phase arithmetic on the last row's
        // bytes (their DMA is clamped to it) and are never stored.  (With `if (row < H)` around it hipcc carried the in-flight ticket
        // through a VECTOR register across the branch -- a copy of the placeholder, which the compiler then could not even lower.)
        float pix[SLX_QUAD];
        {
            const uint32_t *src = ring + lane;
            const uint32_t w0 = src[0 * 64], w1 = src[1 * 64], w2 = src[2 * 64], w3 = src[3 * 64];
            const f32x2 kUp = {0x1p126f, 0x1p126f};
            const F32x2x2 px = wrapped_pix_from_diffs<true>(
                F32x2x2{f32x2{byte_diff_denorm<0>(w0, w2), byte_diff_denorm<1>(w0, w2)} * kUp, f32x2{byte_diff_denorm<2>(w0, w2), byte_diff_denorm<3>(w0, w2)} * kUp},
                F32x2x2{f32x2{byte_diff_denorm<0>(w1, w3), byte_diff_denorm<1>(w1, w3)} * kUp, f32x2{byte_diff_denorm<2>(w1, w3), byte_diff_denorm<3>(w1, w3)} * kUp}, Tf);
            pix[0] = px.a.x;
            pix[1] = px.a.y;
            pix[2] = px.b.x;
            pix[3] = px.b.y;
        }
        // slot 0 is free once it has been read -- and the ticket requested above has arrived with the same wait
        if (ri == 0) {
            fetch_wait(next_raw);
            reset_behind_last(next_raw);
            B = decode(next_raw);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (D.valid && dri == R) request_moves_to(B);
        const unsigned more = uniform_u32(D.valid);                // is there a row after this one?
        if (more) issue_phase();
        // ---- chunk 1: the row's Gray planes.  Behind their DMA: the stores above and the phase DMA just issued
        if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (s > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH + NZ) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPH) : "memory");
        uint32_t gw[NGR];
        {
            const uint32_t *src = ring + ROW_DW + lane;
#pragma unroll
            for (int k = 0; k < NGR; k++) gw[k] = src[k * 64];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // slot 1 is free once it has been read
        if (more) issue_gray();
        __builtin_amdgcn_s_setprio(0);
        if (row < H) {
            // a3 + the bit-pack of a4 on the whole quad (slx_strip_kernel)
            uint32_t acc = 0u;
#pragma unroll
            for (int b = 0; b < GB; b++) acc = swar_push_bit7(acc, swar_ge_u8_bit7(gw[2 * b + 1], gw[2 * b]));
            uint32_t code4 = swar_finish_code(acc, GB);
            if (p.std_gray) code4 = swar_gray_to_binary_u8(code4);   // lut[gray] = bin is the reflected code's inverse (one uniform branch)
            int bin[SLX_QUAD];
#pragma unroll
            for (int jx = 0; jx < SLX_QUAD; jx++) bin[jx] = (int)((code4 >> (8 * jx)) & 0xffu);
            if (!p.std_gray) {
#pragma unroll
                for (int jx = 0; jx < SLX_QUAD; jx++) bin[jx] = (int)p.lut[bin[jx]];
            }
            const double Sd = (double)p.gray_stripe;
            const float q25 = 0.25f * Tf, q75 = 0.75f * Tf, hT = 0.5f * Tf;
            const double vc = (double)((int)row + p.row_offset) - p.cy;
            const double vf = vc * p.fu;
            const double tvC = vf * p.P01, tvD = vf * p.P21;
            double z[SLX_QUAD];
#pragma unroll
            for (int jx = 0; jx < SLX_QUAD; jx++) {
                const double Uv = merge_gray_phase(bin[jx], pix[jx], q25, q75, Tf, hT, Sd);   // a5, R/CCalculation.cpp:570-587
                const double cC = (aC[jx] + tvC) + kK1;
                const double cD = (aD[jx] + tvD) + kK2;
                z[jx] = tri_depth<true>(Uv, cC, cD, kcA, kcB, kfmin, kfmax, true);
            }
            stage[2 * lane + 0] = vec2{z[0], z[1]};
            stage[2 * lane + 1] = vec2{z[2], z[3]};
        }
        // this row's stores go out at the top of the next step
        prsrc = zrsrc;
        pend_off[0] = out_boff[0];
        pend_off[1] = out_boff[1];
        __builtin_amdgcn_wave_barrier();
        // next row of the item, or the next item
        ri = uniform_u32(ri + 1u);
        if (ri == R) {
            if (!B.valid) break;
            A = B;
            ri = 0;
            row = A.row_base + sub;
            zrsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.z + (size_t)A.set * p.out_set_stride), 0, H * W * 8u, 0x00020000);
            out_boff[0] = out_lane[0] + A.row_base * W * 8u;
            out_boff[1] = out_lane[1] + A.row_base * W * 8u;
        } else {
            row += il;
            out_boff[0] += out_step;
            out_boff[1] += out_step;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + k * 64 + lane);
        __builtin_amdgcn_raw_buffer_store_b128(v, prsrc, pend_off[k], 0, 2 /* nt */);
    }
}

// ------------------------------------------------------------------------------------------
// The reference's decoder objects on the strip path: what CDecodePhase::Decode (R/CDecodePhase.cpp:83-96 -> CountResult :48-80:
// four 8-bit planes in, the wrapped phase in projector pixels out as CV_64FC1) and CDecodeGray::Decode (R/CDecodeGray.cpp:108-139
// -> Grey2Bin :150-176, CountResult :179-204: 2 G planes in, the stripe's left edge out as CV_64FC1) compute, for a host loop that
// keeps the reference's two-decoder structure instead of the fused call.  Same work items, same
// HBM -> LDS DMA ring (one chunk = a row of the decoder's planes, two chunks ahead, counted waits), same staged lane-contiguous
// nontemporal stores as slx_strip_kernel above; the arithmetic is that kernel's wrapped_pix_from_diffs / SWAR Gray pack.
//   MODE = SLX_MODE_PHASE_ONLY: 4 planes (the reference's 4 steps), 4 + 8 = 12 B/px
//   MODE = SLX_MODE_GRAY_ONLY:  12 planes (6 bits: the reference's count), reflected-code table, 12 + 8 = 20 B/px
// Everything else (other step / bit counts, a table that is not the reflected code, ragged widths, unequally spaced planes) stays
// with slx_fused_kernel (slx_strip_eligible).
template <int MODE>
__global__ __launch_bounds__(256) void slx_decoder_strip_kernel(const SlxKParams p)
{
    constexpr bool PHASE = MODE == SLX_MODE_PHASE_ONLY;
    constexpr int GB = 6;
    constexpr int NP = PHASE ? 4 : 2 * GB;          // planes of a row = one ring chunk
    constexpr unsigned ROW_DW = NP * 64;            // one ring slot in LDS, dwords per wave
    constexpr int NZ = 2;                           // stores per row
    typedef double vec2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
    const unsigned t = threadIdx.x;
    const unsigned lane = t & 63u;
    const unsigned wave_in_wg = __builtin_amdgcn_readfirstlane(t >> 6);
    uint32_t *ring = lds_raw + wave_in_wg * (2u * ROW_DW + 512u);
    vec2 *stage = reinterpret_cast<vec2 *>(ring + 2u * ROW_DW);
    // tiers of items, as in slx_strip_kernel
    unsigned wg = blockIdx.x;
    unsigned RB = p.tier_rows[0], items_per_set = p.tier_items_per_set[0], region_row0 = 0, tier_items = p.tier_items[0];
#pragma unroll
    for (int tr = 1; tr < SLX_MAX_TIERS; tr++) {
        if (tr < (int)p.n_tiers && blockIdx.x >= p.tier_first_wg[tr]) {
            wg = blockIdx.x - p.tier_first_wg[tr];
            RB = p.tier_rows[tr];
            items_per_set = p.tier_items_per_set[tr];
            region_row0 = p.tier_row0[tr];
            tier_items = p.tier_items[tr];
        }
    }
    const unsigned item = wg * (blockDim.x >> 6) + wave_in_wg;
    if (item >= tier_items) return;
    const unsigned W = (unsigned)p.width, H = (unsigned)p.height;
    const unsigned row_stride = (unsigned)p.row_stride;
    const unsigned step_rows = p.interleave;
    bool lane_valid;
    const StripPos pos = strip_locate<false>(p, item, items_per_set, RB, region_row0, lane_valid);
    // the plan puts the decoder's planes (phase or Gray) behind plane_base / phase_first / phase_step / phase_set_stride
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.plane_base + (size_t)pos.set * p.phase_set_stride), 0, 0xFFFFFFFFu, 0x00020000);
    const unsigned dma_step = step_rows * row_stride;
    const unsigned dma_last = (H - 1u) * row_stride + pos.cq * SLX_QUAD;   // rows past the tile: harmless re-read of the last row
    unsigned dma_off = pos.row * row_stride + pos.cq * SLX_QUAD;
    auto next_plane = [](unsigned &so, unsigned step) { asm volatile("s_add_u32 %0, %0, %1" : "+s"(so) : "s"(step) : "scc"); };
    auto issue_chunk = [&](unsigned slot) {
        const unsigned voff = dma_off < dma_last ? dma_off : dma_last;
        dma_off += dma_step;
        uint32_t *dst = ring + slot * ROW_DW;
        unsigned so = p.phase_first;
        if (p.dma_imm) {
            static_for<NP>([&](auto k) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)dst, 4, voff, so, decltype(k)::value * 256, 2 /* nt */);
                if (decltype(k)::value + 1 < NP) next_plane(so, p.phase_step - 256u);
            });
        } else {
#pragma unroll
            for (int k = 0; k < NP; k++) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(dst + k * 64), 4, voff, so, 0, 2 /* nt */);
                if (k + 1 < NP) next_plane(so, p.phase_step);
            }
        }
    };
    double *oset = (PHASE ? p.pix : p.gray_out) + (size_t)pos.set * p.out_set_stride;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(oset, 0, H * W * 8u, 0x00020000);
    unsigned out_boff[2], out_bstep[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const bool ok = pos.out_row[k] != 0xFFFFFFFFu;
        out_boff[k] = ok ? pos.out_off[k] * 8u : 0xFFFFFFF0u;
        out_bstep[k] = ok ? step_rows * W * 8u : 0u;
    }
    auto flush_row = [&]() {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + k * 64 + lane);
            __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, out_boff[k], 0, 2 /* nt */);
            out_boff[k] += out_bstep[k];
        }
    };
    const float Tf = (float)p.period[0];
    const double Sd = (double)p.gray_stripe;

    issue_chunk(0);
    if (RB > 1) issue_chunk(1);
    for (unsigned i = 0; i < RB; i++) {
        const unsigned row = pos.row + i * step_rows;
        const unsigned slot = i & 1u;
        // counted waits (vmcnt retires in issue order, loads and stores alike): L(i) | Z(i-2) L(i+1) | wait -- see slx_strip_kernel
        if (i + 1 >= RB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (i >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + NZ) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        if (i > 0) flush_row();                                         // last row's stores, one step late
        uint32_t w[NP];
        if (row < H) {
            const uint32_t *src = ring + slot * ROW_DW + lane;
#pragma unroll
            for (int k = 0; k < NP; k++) w[k] = src[k * 64];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slot is free once it has been read
        }
        if (i + 2 < RB) issue_chunk(slot);
        if (row < H) {
            double o[SLX_QUAD];
            if constexpr (PHASE) {
                // a1 + a2: R/CDecodePhase.cpp:59-75, the quad's two pixel pairs in lockstep (wrapped_pix_from_diffs)
                const f32x2 kUp = {0x1p126f, 0x1p126f};
                const F32x2x2 px = wrapped_pix_from_diffs<true>(
                    F32x2x2{f32x2{byte_diff_denorm<0>(w[0], w[2]), byte_diff_denorm<1>(w[0], w[2])} * kUp, f32x2{byte_diff_denorm<2>(w[0], w[2]), byte_diff_denorm<3>(w[0], w[2])} * kUp},
                    F32x2x2{f32x2{byte_diff_denorm<0>(w[1], w[3]), byte_diff_denorm<1>(w[1], w[3])} * kUp, f32x2{byte_diff_denorm<2>(w[1], w[3]), byte_diff_denorm<3>(w[1], w[3])} * kUp}, Tf);
                o[0] = (double)px.a.x;                                  // R/CDecodePhase.cpp:75
                o[1] = (double)px.a.y;
                o[2] = (double)px.b.x;
                o[3] = (double)px.b.y;
            } else {
                // a3 + a4: bit = pattern > inverse (ties -> 0), pair 0 = LSB, lut[gray] = bin (the reflected code's inverse), x stripe
                uint32_t acc = 0u;
#pragma unroll
                for (int b = 0; b < GB; b++) acc = swar_push_bit7(acc, swar_ge_u8_bit7(w[2 * b + 1], w[2 * b]));
                const uint32_t bin4 = swar_gray_to_binary_u8(swar_finish_code(acc, GB));
#pragma unroll
                for (int j = 0; j < SLX_QUAD; j++) o[j] = (double)(int)((bin4 >> (8 * j)) & 0xffu) * Sd;   // R/CDecodeGray.cpp:200
            }
            stage[2 * lane + 0] = vec2{o[0], o[1]};
            stage[2 * lane + 1] = vec2{o[2], o[3]};
        }
        __builtin_amdgcn_wave_barrier();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    flush_row();
}

// ------------------------------------------------------------------------------------------
// Point cloud (CCalculation::Result, R/CCalculation.cpp:323-357): the reference walks u outer / v inner and
// writes "x y z" for every depth inside the FOV.  Here the depth map is cut into 64 x 64 tiles; the cloud's order
// (column, then row) is the order of the entries (column u, row block rb = v / 64) laid out as u * RB + rb, so a strip of
// 64 columns (tile column bx) owns the contiguous entries [64 bx RB, 64 (bx + 1) RB).  Two launches, no scan kernel:
//   count:  entry (u, rb) = kept depths of column u in rows [64 rb, 64 rb + 64) -- rows read coalesced -- and the tile's
//           total, tiles[bx RB + rb]
//   write:  a tile's workgroup finds its 64 offsets itself: the tile totals of the strips before it (a few hundred
//           values, every workgroup sums them), the column totals of its own strip before each column, and the column's
//           entries above its row block.  The tile then goes through LDS (coalesced rows in, columns out); a wave takes a
//           column, lane = row, the kept lanes' rank (ballot + popcount below the lane) places the point in the column's
//           run, the run is packed in LDS and leaves as contiguous doubles, 512 bytes per store.
// (A single-workgroup scan kernel between the two cost 14-18 us whatever its shape: one workgroup fetching cold code.)
constexpr int kCloudTile = 64;

__device__ __forceinline__ bool cloud_keep(double zz, double fov_min, double fov_max)
{
    return !((zz < fov_min) || (zz > fov_max));                     // the reference's `continue` test, negated
}

__global__ __launch_bounds__(256) void slx_cloud_count_kernel(const double *z, unsigned *counts, unsigned *tiles, int W, int H, int RB, double fov_min,
                                                             double fov_max)
{
    __shared__ unsigned part[4][kCloudTile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blockIdx.x * kCloudTile + lane, rb = blockIdx.y;
    const int uu = u < W ? u : W - 1;
    double zz[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {                                  // all 16 rows in flight: clamped addresses, masked below
        const int v = rb * kCloudTile + wave * 16 + k;
        zz[k] = z[(size_t)(v < H ? v : H - 1) * W + uu];
    }
    unsigned n = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int v = rb * kCloudTile + wave * 16 + k;
        n += (v < H && u < W && cloud_keep(zz[k], fov_min, fov_max)) ? 1u : 0u;
    }
    part[wave][lane] = n;
    __syncthreads();
    if (wave == 0) {
        unsigned col = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        if (u < W) counts[(size_t)u * RB + rb] = col;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) col += __shfl_xor(col, d);
        if (lane == 0) tiles[(size_t)blockIdx.x * RB + rb] = col;
    }
}

// FAST: x and y by slx_div_item_const (fu, fv checked on the host to sit inside its range) -- the same quotient bits.
// total_dev / total_host: where workgroup (0, 0) leaves the number of points (a device word and a pinned host word).
template <bool FAST>
__global__ __launch_bounds__(256) void slx_cloud_write_kernel(const double *z, const unsigned *counts, const unsigned *tiles, double *xyz,
                                                             unsigned *total_dev, unsigned *total_host, int W, int H, int GX, int RB, int row_offset,
                                                             double fov_min, double fov_max, double cx, double cy, double fu, double fv)
{
    __shared__ double tile[kCloudTile][kCloudTile + 1];
    __shared__ double run[4][3 * kCloudTile];                       // a column's records, packed, one wave each
    __shared__ unsigned col_part[4][kCloudTile], above_part[4][kCloudTile], col_off[kCloudTile], red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u0 = blockIdx.x * kCloudTile, rb = blockIdx.y, v0 = rb * kCloudTile;
    const int u = u0 + lane, uu = u < W ? u : W - 1;
    double zz[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {                                  // rows in, 512 contiguous bytes per wave, all in flight
        const int v = v0 + wave * 16 + k;
        zz[k] = z[(size_t)(v < H ? v : H - 1) * W + uu];
    }
    // offsets, while those loads fly.  (1) points before this strip, and all points:
    const int n_tiles = GX * RB, before = (int)blockIdx.x * RB;
    unsigned s_before = 0, s_all = 0;
    for (int i = threadIdx.x; i < n_tiles; i += 256) {
        const unsigned t = tiles[i];
        s_all += t;
        s_before += i < before ? t : 0u;
    }
    // (2) this strip: column lane's total and its entries above row block rb, a quarter of the row blocks per wave
    unsigned c_all = 0, c_above = 0;
    if (u < W) {
        const unsigned *col = counts + (size_t)u * RB;
        for (int r = wave; r < RB; r += 4) {
            const unsigned t = col[r];
            c_all += t;
            c_above += r < rb ? t : 0u;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s_before += __shfl_xor(s_before, d);
        s_all += __shfl_xor(s_all, d);
    }
    if (lane == 0) red[wave] = s_before, red[4 + wave] = s_all;
    col_part[wave][lane] = c_all;
    above_part[wave][lane] = c_above;
#pragma unroll
    for (int k = 0; k < 16; k++) tile[wave * 16 + k][lane] = zz[k];  // rows / columns past the image: masked by v < H, u < W below
    __syncthreads();
    if (wave == 0) {
        const unsigned mine = col_part[0][lane] + col_part[1][lane] + col_part[2][lane] + col_part[3][lane];
        unsigned incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        col_off[lane] = (red[0] + red[1] + red[2] + red[3]) + (incl - mine) + (above_part[0][lane] + above_part[1][lane] + above_part[2][lane] + above_part[3][lane]);
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
            const unsigned total = red[4] + red[5] + red[6] + red[7];
            *total_dev = total;
            if (total_host) *total_host = total;
        }
    }
    __syncthreads();
    if (!xyz) return;                                               // count only: the caller wanted the number of points
    const int v = v0 + lane;
    const double vc = (double)(v + row_offset) - cy;                // R/CCalculation.cpp:763
    const double ru = FAST ? slx_refined_rcp_f64(fu) : 0.0, rv = FAST ? slx_refined_rcp_f64(fv) : 0.0;
    double *mine = run[wave];
    for (int k = 0; k < 16; k++) {
        const int c = wave * 16 + k, uc_i = u0 + c;
        if (uc_i >= W) break;                                       // uniform over the wave
        const double zc = tile[lane][c];
        const bool keep = v < H && cloud_keep(zc, fov_min, fov_max);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) {
            const unsigned rank = (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            const double uc = (double)uc_i - cx;                    // :762
            mine[3 * rank + 0] = FAST ? slx_div_item_const(zc * uc, fu, ru) : zc * uc / fu;   // :766
            mine[3 * rank + 1] = FAST ? slx_div_item_const(zc * vc, fv, rv) : zc * vc / fv;   // :767
            mine[3 * rank + 2] = zc;
        }
        // the column's records leave as one contiguous run of doubles, 512 bytes per store (a wave's LDS accesses
        // execute in order: the reads below see the writes above, and the next column's writes come after these reads)
        __builtin_amdgcn_wave_barrier();
        const unsigned words = 3u * (unsigned)__builtin_popcountll(m);
        double *dst = xyz + (size_t)col_off[c] * 3;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const unsigned w = (unsigned)lane + 64u * j;
            if (w < words) dst[w] = mine[w];
        }
        __builtin_amdgcn_wave_barrier();
    }
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

template <int MODE, int GB, int NS = 4>
kernel_fn pick_strip(int F, bool aux)
{
    switch (F) {
    case 1: return aux ? slx_strip_kernel<MODE, 1, GB, NS, true> : slx_strip_kernel<MODE, 1, GB, NS, false>;
    case 2: return aux ? slx_strip_kernel<MODE, 2, GB, NS, true> : slx_strip_kernel<MODE, 2, GB, NS, false>;
    case 3: return aux ? slx_strip_kernel<MODE, 3, GB, NS, true> : slx_strip_kernel<MODE, 3, GB, NS, false>;
    case 4: return aux ? slx_strip_kernel<MODE, 4, GB, NS, true> : slx_strip_kernel<MODE, 4, GB, NS, false>;
    }
    return nullptr;
}

}  // namespace

int slx_launch_cloud_count(const SlxKParams &kp, const double *z, unsigned *counts, unsigned *tiles, void *stream)
{
    const int RB = (kp.height + kCloudTile - 1) / kCloudTile;
    hipLaunchKernelGGL(slx_cloud_count_kernel, dim3((kp.width + kCloudTile - 1) / kCloudTile, RB), dim3(256), 0, (hipStream_t)stream, z, counts, tiles,
                       kp.width, kp.height, RB, kp.fov_min, kp.fov_max);
    return (int)hipGetLastError();
}

int slx_launch_cloud_write(const SlxKParams &kp, const double *z, const unsigned *counts, const unsigned *tiles, double *xyz, unsigned *total_dev,
                           unsigned *total_host, void *stream)
{
    const int RB = (kp.height + kCloudTile - 1) / kCloudTile;
    auto in_range = [](double d) { return __builtin_fabs(d) > 0x1p-90 && __builtin_fabs(d) < 0x1p90; };
    auto fn = in_range(kp.fu) && in_range(kp.fv) ? slx_cloud_write_kernel<true> : slx_cloud_write_kernel<false>;
    const int GX = (kp.width + kCloudTile - 1) / kCloudTile;
    // without a target only the number of points is wanted: workgroup (0, 0) alone sums the tile totals
    hipLaunchKernelGGL(fn, xyz ? dim3(GX, RB) : dim3(1, 1), dim3(256), 0, (hipStream_t)stream, z, counts, tiles, xyz, total_dev,
                       total_host, kp.width, kp.height, GX, RB, kp.row_offset, kp.fov_min, kp.fov_max, kp.cx, kp.cy, kp.fu, kp.fv);
    return (int)hipGetLastError();
}

// Launches a plan of slx_plan_launch (slx_plan.cpp: kernel choice, work items, grid -- host arithmetic only).
int slx_launch_fused(const SlxKParams &kp_in, int mode, bool aux, int n_sets, int variant, void *stream, const SlxTuning *tune, SlxStreamState *st)
{
    SlxLaunchPlan plan;
    SlxKParams kq = kp_in;
    kq.sq_counters = (st && variant != SLX_VARIANT_GENERIC && variant != SLX_VARIANT_GENERIC_FAST) ? st->counters : nullptr;
    if (slx_plan_launch(kq, mode, aux, n_sets, variant, tune, &plan) != 0) return (int)hipErrorInvalidValue;
    SlxKParams &kp = plan.kp;
    kernel_fn fn;
    if (plan.stream) {
        // the kernels leave their counters at zero (the wave that draws a queue's last ticket resets it): the words are zeroed here only
        // before their first use and after a launch that failed (st->key == 0)
        if (st->key == 0) {
            const hipError_t e = hipMemsetAsync(st->counters, 0, (size_t)SLX_STREAM_MAX_QUEUES * 32u * sizeof(unsigned), (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
            st->key = 1;
        }
        if (plan.stream == 2) {
            fn = slx_gstream_kernel;
        } else {
            switch (kp.n_freq) {
            case 1: fn = slx_stream_kernel<1>; break;
            case 2: fn = slx_stream_kernel<2>; break;
            case 3: fn = slx_stream_kernel<3>; break;
            default: fn = slx_stream_kernel<4>; break;
            }
        }
    } else if (!plan.strip) {
        fn = pick(mode, kp.n_freq, kp.n_steps == 4, aux);
    } else if (mode == SLX_MODE_PHASE_ONLY || mode == SLX_MODE_GRAY_ONLY) {
        fn = mode == SLX_MODE_PHASE_ONLY ? slx_decoder_strip_kernel<SLX_MODE_PHASE_ONLY> : slx_decoder_strip_kernel<SLX_MODE_GRAY_ONLY>;
    } else {
        const int gb = plan.gray_ring_bits;
        fn = mode == SLX_MODE_MULTIFREQ ? (kp.n_steps == 8 ? pick_strip<SLX_MODE_MULTIFREQ, 0, 8>(kp.n_freq, aux) : pick_strip<SLX_MODE_MULTIFREQ, 0>(kp.n_freq, aux))
             : mode == SLX_MODE_MULTIFREQ_GRAYMASK
                 ? (gb ? pick_strip<SLX_MODE_MULTIFREQ_GRAYMASK, 6>(kp.n_freq, aux) : pick_strip<SLX_MODE_MULTIFREQ_GRAYMASK, 0>(kp.n_freq, aux))
                 : (gb ? pick_strip<SLX_MODE_GRAY_PHASE, 6>(1, aux) : pick_strip<SLX_MODE_GRAY_PHASE, 0>(1, aux));
    }
    if (!fn) return (int)hipErrorInvalidValue;
    if (st) {
        st->last_kind = plan.stream == 2 ? 5 : plan.stream ? 3 : !plan.strip ? 1 : (mode == SLX_MODE_PHASE_ONLY || mode == SLX_MODE_GRAY_ONLY) ? 4 : 2;
        st->last_rows = plan.stream ? (int)kp.sq_rows : plan.strip ? (int)kp.tier_rows[0] : 0;
        st->last_weave = plan.strip ? (int)kp.interleave : 0;
        st->last_mode = mode;
        st->last_freq = mode == SLX_MODE_GRAY_ONLY ? 0 : (mode == SLX_MODE_PHASE_ONLY || mode == SLX_MODE_GRAY_PHASE) ? 1 : kp.n_freq;
        st->last_gray_ring_bits = plan.strip && !plan.stream ? plan.gray_ring_bits : 0;
        st->last_steps = kp.n_steps;
        st->last_aux = aux ? 1 : 0;
    }
    // operand shapes were validated by the caller (slx_api.cpp: check_launch_shapes), slx_strip_eligible and the plan
    hipLaunchKernelGGL(fn, dim3(plan.grid_x, plan.grid_y, 1), dim3(plan.block, 1, 1), plan.lds_bytes, (hipStream_t)stream, kp);
    return (int)hipGetLastError();
}
