// dtl_math.h -- single-precision exp / log / sincos with a fixed IEEE operation sequence.
//
// The receive chain's soft outputs feed an iterative decoder whose non-converging codewords amplify
// last-bit differences; libm-style "<= 1 ulp, unspecified bits" functions (or v_exp_f32/v_log_f32)
// would make the soft bits irreproducible against a CPU run.  These routines use only IEEE single
// operations in a fixed order (Cody-Waite reduction, fdlibm-class polynomials, no FMA contraction --
// the including file sets `#pragma clang fp contract(off)`), so a CPU that executes the same sequence
// (the test oracle does) obtains bit-identical results.  Accuracy <= 2 ulp.
#pragma once

#include <hip/hip_runtime.h>

namespace dtl {

__device__ __forceinline__ float dm_exp(float x)
{
    if (x != x) return x;
    if (x > 88.72283905f) return __int_as_float(0x7f800000);
    if (x < -103.97208f) return 0.0f;
    const float kf = rintf(x * 1.44269504f);
    const int k = int(kf);
    float r = x - kf * 0.693145752f;
    r = r - kf * 1.42860677e-06f;
    float p = 1.9841270e-4f;
    p = p * r + 1.3888889e-3f;
    p = p * r + 8.3333338e-3f;
    p = p * r + 4.1666668e-2f;
    p = p * r + 1.6666667e-1f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    // p * 2^k with a single rounding (v_ldexp_f32); bit-identical to the CPU oracle's two-step
    // (p * 2^(k/2)) * 2^(k - k/2), whose first product is exact (tools/math_check.hip: all 2^32 inputs)
    return ldexpf(p, k);
}

__device__ __forceinline__ float dm_log(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return __int_as_float(0x7fc00000);
    if (x == 0.0f) return __int_as_float(0xff800000);
    if (x == __int_as_float(0x7f800000)) return x;
    int e = 0;
    if (x < 1.17549435e-38f) { x = x * 33554432.0f; e = -25; }
    unsigned ix = unsigned(__float_as_int(x));
    e += int(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) | 0x3f800000u;
    float m = __int_as_float(int(ix));
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float R = z * (0.66666662693f + z * (0.40000972152f + z * (0.28498786688f + z * 0.24279078841f)));
    const float hfsq = 0.5f * f * f;
    const float ef = float(e);
    return ef * 6.9313812256e-01f + (f - (hfsq - (s * (hfsq + R) + ef * 9.0580006145e-06f)));
}

__device__ __forceinline__ void dm_sincos(float x, float* sn, float* cs)
{
    if (!(x == x) || x == __int_as_float(0x7f800000) || x == __int_as_float(0xff800000)) {
        *sn = *cs = __int_as_float(0x7fc00000);
        return;
    }
    double xd = double(x);
    if (xd >= 3.14159265358979323846 || xd < -3.14159265358979323846) {
        const double kk = floor((xd + 3.14159265358979323846) * (1.0 / 6.28318530717958647692));
        xd = xd - 6.28318530717958647692 * kk;
    }
    const float xr = float(xd);
    const float qf = rintf(xr * 0.636619772f);
    const int q = int(qf);
    float r = xr - qf * 1.57079601e+00f;
    r = r - qf * 3.13916473e-07f;
    const float z = r * r;
    const float sp = r + r * z * (-1.6666667163e-01f + z * (8.3333337680e-03f + z * (-1.9841270114e-04f + z * 2.7557314297e-06f)));
    const float cp = 1.0f - 0.5f * z + z * z * (4.1666667908e-02f + z * (-1.3888889225e-03f + z * (2.4801587642e-05f + z * -2.7557314297e-07f)));
    switch (q & 3) {
    case 0: *sn = sp; *cs = cp; break;
    case 1: *sn = cp; *cs = -sp; break;
    case 2: *sn = -sp; *cs = -cp; break;
    default: *sn = -cp; *cs = sp; break;
    }
}

} // namespace dtl
