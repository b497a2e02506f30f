// slx_device.h -- device helpers shared by the decode and the tracker kernels.
#ifndef SLX_DEVICE_H
#define SLX_DEVICE_H

#include <hip/hip_runtime.h>

// n / d for a divisor that is constant over a work item (a7's second pass: x = (z uc) / fu, y = (z vc) / fv,
// R/CCalculation.cpp:766-767).  r = slx_refined_rcp_f64(d): v_rcp_f64 + two Newton steps, the first half of the IEEE division
// sequence hipcc emits, formed once per item; the quotient then takes that sequence's product + residual correction.
// Bit-identical to the IEEE division whenever the sequence needs no scaling (host-checked: 2^-90 < |d| < 2^90):
//  * a zero numerator gives a zero of the quotient's sign: the correction step would return +0, so the sign is put back;
//  * a NaN (overflowing n r, NaN numerator) or a quotient so small that the residual may be inexact takes the literal division.
__device__ __forceinline__ double slx_refined_rcp_f64(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    return __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
}

__device__ __forceinline__ double slx_div_item_const(double n, double d, double r)
{
    const double q = n * r;
    double o = __builtin_fma(__builtin_fma(-d, q, n), r, q);
    const unsigned long long sign = (__builtin_bit_cast(unsigned long long, n) ^ __builtin_bit_cast(unsigned long long, d)) & 0x8000000000000000ull;
    o = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, o) | sign);     // a no-op unless o is a zero (or a NaN)
    if (__builtin_expect((o != o) | ((__builtin_fabs(o) < 0x1p-900) & (o != 0.0)), 0)) o = n / d;
    return o;
}

#endif
