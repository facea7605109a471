// dtl_math.h -- single-precision exp / log / sincos with a fixed IEEE operation sequence.
//
// The receive chain's soft outputs feed an iterative decoder whose non-converging codewords amplify
// last-bit differences; libm-style "<= 1 ulp, unspecified bits" functions (or v_exp_f32/v_log_f32)
// would make the soft bits irreproducible against a CPU run.  These routines use only IEEE single
// operations in a fixed order (Cody-Waite reduction, fdlibm-class polynomials in Horner form with the
// fused multiply-adds written out; nothing is left to contraction -- the including file sets
// `#pragma clang fp contract(off)`), so a CPU that executes the same sequence
// (the test oracle does) obtains bit-identical results.  Accuracy <= 2 ulp.
#pragma once

#include <hip/hip_runtime.h>

namespace dtl {

// Both functions are written branch-free: the main path runs for every input and the special cases are
// selected at the end (v_cndmask), because a wave that demaps 64 symbols would otherwise pay ~150 scalar
// branches per symbol.  Results are those of the straightforward early-return forms (the CPU oracle's).
// NONPOS: the caller guarantees x <= 0 and not NaN (then the overflow and NaN selects do nothing and are left out)
template <bool NONPOS = false>
__device__ __forceinline__ float dm_exp(float x)
{
    const float kf = rintf(x * 1.44269504f);
    const int k = int(kf);
    float r = __builtin_fmaf(-kf, 0.693145752f, x);
    r = __builtin_fmaf(-kf, 1.42860677e-06f, r);
    float p = 1.9841270e-4f;
    p = __builtin_fmaf(p, r, 1.3888889e-3f);
    p = __builtin_fmaf(p, r, 8.3333338e-3f);
    p = __builtin_fmaf(p, r, 4.1666668e-2f);
    p = __builtin_fmaf(p, r, 1.6666667e-1f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    // p * 2^k with a single rounding (v_ldexp_f32); bit-identical to the CPU oracle's two-step
    // (p * 2^(k/2)) * 2^(k - k/2), whose first product is exact (tools/math_check.hip: all 2^32 inputs)
    float res = ldexpf(p, k);
    if constexpr (NONPOS) return (x < -103.97208f) ? 0.0f : res;
    res = (x > 88.72283905f) ? __int_as_float(0x7f800000) : res;
    res = (x < -103.97208f) ? 0.0f : res;
    return (x != x) ? x : res;
}

// Two values at a time (v_pk_mul_f32 / v_pk_fma_f32: one instruction, two IEEE operations): the sequences of
// dm_neg_div_by<true> and dm_exp<true> applied to both components -- for waves whose operands are ordinary (the demapper's
// ORD form), where neither has a special case left except the exponential's underflow clamp.
typedef float dm_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ dm_f2 dm_fma2(dm_f2 a, dm_f2 b, dm_f2 c) { return __builtin_elementwise_fma(a, b, c); }

// exp(-a / b) for K pairs at once, the K chains written side by side (every packed instruction then has K - 1 independent
// ones between itself and its consumer: no wait states between dependent packed operations)
template <int K>
__device__ __forceinline__ void dm_exp_neg_div_by_pairs(const dm_f2 (&a)[K], float b, float r, dm_f2 (&out)[K])
{
    const dm_f2 bb = { b, b }, rr = { r, r };
    dm_f2 q[K], e[K], x[K], kf[K], p[K];
#pragma unroll
    for (int i = 0; i < K; i++) q[i] = a[i] * rr;
#pragma unroll
    for (int i = 0; i < K; i++) e[i] = dm_fma2(-bb, q[i], a[i]);
#pragma unroll
    for (int i = 0; i < K; i++) q[i] = dm_fma2(e[i], rr, q[i]);
#pragma unroll
    for (int i = 0; i < K; i++) e[i] = dm_fma2(-bb, q[i], a[i]);
#pragma unroll
    for (int i = 0; i < K; i++) x[i] = -dm_fma2(e[i], rr, q[i]);
#pragma unroll
    for (int i = 0; i < K; i++) {
        const dm_f2 t = x[i] * dm_f2{ 1.44269504f, 1.44269504f };
        kf[i] = dm_f2{ rintf(t.x), rintf(t.y) };
    }
#pragma unroll
    for (int i = 0; i < K; i++) e[i] = dm_fma2(-kf[i], dm_f2{ 0.693145752f, 0.693145752f }, x[i]);
#pragma unroll
    for (int i = 0; i < K; i++) e[i] = dm_fma2(-kf[i], dm_f2{ 1.42860677e-06f, 1.42860677e-06f }, e[i]);
#pragma unroll
    for (int i = 0; i < K; i++) p[i] = dm_fma2(dm_f2{ 1.9841270e-4f, 1.9841270e-4f }, e[i], dm_f2{ 1.3888889e-3f, 1.3888889e-3f });
#pragma unroll
    for (int i = 0; i < K; i++) p[i] = dm_fma2(p[i], e[i], dm_f2{ 8.3333338e-3f, 8.3333338e-3f });
#pragma unroll
    for (int i = 0; i < K; i++) p[i] = dm_fma2(p[i], e[i], dm_f2{ 4.1666668e-2f, 4.1666668e-2f });
#pragma unroll
    for (int i = 0; i < K; i++) p[i] = dm_fma2(p[i], e[i], dm_f2{ 1.6666667e-1f, 1.6666667e-1f });
#pragma unroll
    for (int i = 0; i < K; i++) p[i] = dm_fma2(p[i], e[i], dm_f2{ 0.5f, 0.5f });
#pragma unroll
    for (int i = 0; i < K; i++) p[i] = dm_fma2(p[i], e[i], dm_f2{ 1.0f, 1.0f });
#pragma unroll
    for (int i = 0; i < K; i++) p[i] = dm_fma2(p[i], e[i], dm_f2{ 1.0f, 1.0f });
#pragma unroll
    for (int i = 0; i < K; i++) {
        dm_f2 res = { ldexpf(p[i].x, int(kf[i].x)), ldexpf(p[i].y, int(kf[i].y)) };
        res.x = (x[i].x < -103.97208f) ? 0.0f : res.x;
        res.y = (x[i].y < -103.97208f) ? 0.0f : res.y;
        out[i] = res;
    }
}

// exp(x) in double with the same construction (k = rint(x log2e), r = x - k ln2 with fdlibm's hi/lo split, Taylor
// polynomial of degree 13 in Horner form, 2^k applied last); < 1 ulp.  The LDPC decoder's channel likelihood ratios use
// it instead of the library exp, whose last bit is the library's own (ocml and glibc differ at x = 40): the CPU oracle
// executes the same sequence (oracle/orc_math.h orc_exp).
__device__ __forceinline__ double dm_exp64(double x)
{
    const double kd = rint(x * 0x1.71547652b82fep+0);
    double r = __builtin_fma(-kd, 0x1.62e42fee00000p-1, x);
    r = __builtin_fma(-kd, 0x1.a39ef35793c76p-33, r);
    double p = 0x1.6124613a86d09p-33;
    p = __builtin_fma(p, r, 0x1.1eed8eff8d898p-29);
    p = __builtin_fma(p, r, 0x1.ae64567f544e4p-26);
    p = __builtin_fma(p, r, 0x1.27e4fb7789f5cp-22);
    p = __builtin_fma(p, r, 0x1.71de3a556c734p-19);
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-16);
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-13);
    p = __builtin_fma(p, r, 0x1.6c16c16c16c17p-10);
    p = __builtin_fma(p, r, 0x1.1111111111111p-7);
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0x1.0000000000000p-1);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    // |x| <= 746 on this path: |k| <= 1077, the two-step scaling keeps both factors normal; the first product is exact
    const int k = int(fmin(fmax(kd, -1100.0), 1100.0));
    const int k1 = k / 2, k2 = k - k1;
    double res = (p * __longlong_as_double((long long)(k1 + 1023) << 52)) * __longlong_as_double((long long)(k2 + 1023) << 52);
    res = (x > 710.0) ? __longlong_as_double(0x7ff0000000000000LL) : res;
    res = (x < -746.0) ? 0.0 : res;
    return (x != x) ? x : res;
}

// log / log10 / 10^y in double, written out (oracle/orc_math.h orc_log, orc_log10, orc_pow10 execute the same sequence):
// the frame's SNR and noise figures scale every soft bit of the frame, and ocml's log10 / pow differ from the C library's
// in the last bits often enough to show.  Only the frame equaliser's epilogue calls them (once per frame).
__device__ __forceinline__ double dm_log64(double x)
{
    if (x != x) return x;
    if (x < 0.0) return __longlong_as_double(0x7ff8000000000000LL);
    if (x == 0.0) return __longlong_as_double(0xfff0000000000000LL);
    if (x == __longlong_as_double(0x7ff0000000000000LL)) return x;
    int k = 0;
    if (x < 0x1p-1022) { x *= 0x1p54; k = -54; }
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    k += int(b >> 52) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)b);
    if (m > 0x1.6a09e667f3bcdp+0) { m *= 0.5; k += 1; }
    const double f = m - 1.0, s = f / (2.0 + f), z = s * s, w = z * z, dk = double(k);
    const double t1 = w * (0x1.999999997fa04p-2 + w * (0x1.c71c51d8e78afp-3 + w * 0x1.39a09d078c69fp-3));
    const double t2 = z * (0x1.5555555555593p-1 + w * (0x1.2492494229359p-2 + w * (0x1.7466496cb03dep-3 + w * 0x1.2f112df3e5244p-3)));
    const double R = t2 + t1, hfsq = 0.5 * f * f;
    return dk * 0x1.62e42fee00000p-1 - ((hfsq - (s * (hfsq + R) + dk * 0x1.a39ef35793c76p-33)) - f);
}
__device__ __forceinline__ double dm_log10_64(double x) { return dm_log64(x) * 0x1.bcb7b1526e50ep-2; }
__device__ __forceinline__ double dm_pow10_64(double y)
{
    const double p = y * 0x1.26bb1bbb55516p+1;
    const double r = dm_exp64(p);
    if (!(p > -800.0 && p < 800.0) || r == 0.0 || r == __longlong_as_double(0x7ff0000000000000LL)) return r;
    const double e = __builtin_fma(y, 0x1.26bb1bbb55516p+1, -p) + y * -0x1.f48ad494ea3e9p-53;
    return __builtin_fma(r, e, r);
}

// a / b for operands that need none of v_div_scale / v_div_fixup's range handling (no scaling, no
// special values): the refinement sequence the compiler emits for an IEEE single division, minus those
// two steps -- same bits, 8 instructions instead of 11 and no VCC hazard.  Only used by dm_log, where
// a in [-0.2929, 0.4143] is a multiple of 2^-24 and b = 2 + a (exhaustively checked, tools/math_check.hip).
__device__ __forceinline__ float dm_div_inrange(float a, float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, y0, 1.0f);
    const float y1 = __builtin_fmaf(e0, y0, y0);
    const float q0 = a * y1;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y1, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y1, q1);
}

// -a / b for many numerators a >= 0 and one divisor b whose correctly rounded reciprocal r = RN(1 / b) is known:
// the quotient-refinement half of the IEEE division sequence (two residual corrections with fused multiply-adds),
// which rounds correctly once the reciprocal is within an ulp -- here it is exact to half an ulp.  Anything that
// is not an ordinary finite quotient takes the plain division.  Results in the subnormal range may differ in the
// last bit, which the demapper cannot see (exp of both is exactly 1).  tools/math_check.hip compares with -a / b.
// ORDINARY: the caller guarantees a * r < 3e38 and b < 3e38 (the guard does nothing and is left out)
template <bool ORDINARY = false>
__device__ __forceinline__ float dm_neg_div_by(float a, float b, float r)
{
    float q = a * r;
    if constexpr (!ORDINARY) { if (!(q < 3.0e38f) || !(b < 3.0e38f)) return -a / b; }
    float e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, r, q);
    return -q;
}

// a / n in double for a divisor whose correctly rounded reciprocal r = RN(1 / n) comes from a table (the sample
// counts of the SNR estimator): same refinement, same result as the IEEE division for every finite quotient that
// is not subnormal -- the estimator's operands (differences of values near 1, small integer counts) never are;
// anything else takes the plain division.
__device__ __forceinline__ double dm_div_by_count(double a, double n, double r)
{
    double q = a * r;
    if (!(fabs(q) < 1.0e300) || (q != 0.0 && fabs(q) < 1.0e-290)) return a / n;
    double e = __builtin_fma(-n, q, a);
    q = __builtin_fma(e, r, q);
    e = __builtin_fma(-n, q, a);
    return __builtin_fma(e, r, q);
}

// NORMAL: the caller guarantees 2^-126 <= x0 < inf (no rescaling of tiny arguments, no special results)
template <bool NORMAL = false>
__device__ __forceinline__ float dm_log(float x0)
{
    const bool tiny = !NORMAL && x0 < 1.17549435e-38f;
    const float x = tiny ? x0 * 33554432.0f : x0;
    int e = tiny ? -25 : 0;
    unsigned ix = unsigned(__float_as_int(x));
    e += int(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) | 0x3f800000u;
    float m = __int_as_float(int(ix));
    const bool big = m > 1.41421354f;
    m = big ? m * 0.5f : m;
    e += big ? 1 : 0;
    const float f = m - 1.0f;
    const float s = dm_div_inrange(f, 2.0f + f);
    const float z = s * s;
    const float R = z * __builtin_fmaf(z, __builtin_fmaf(z, __builtin_fmaf(z, 0.24279078841f, 0.28498786688f), 0.40000972152f), 0.66666662693f);
    const float hfsq = 0.5f * f * f;
    const float ef = float(e);
    float res = __builtin_fmaf(ef, 6.9313812256e-01f, f - (hfsq - __builtin_fmaf(s, hfsq + R, ef * 9.0580006145e-06f)));
    if constexpr (NORMAL) return res;
    res = (x0 == __int_as_float(0x7f800000)) ? x0 : res;
    res = (x0 == 0.0f) ? __int_as_float(0xff800000) : res;
    res = (x0 < 0.0f) ? __int_as_float(0x7fc00000) : res;
    return (x0 != x0) ? x0 : res;
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
    float r = __builtin_fmaf(-qf, 1.57079601e+00f, xr);
    r = __builtin_fmaf(-qf, 3.13916473e-07f, r);
    const float z = r * r;
    const float ps = __builtin_fmaf(z, __builtin_fmaf(z, __builtin_fmaf(z, 2.7557314297e-06f, -1.9841270114e-04f), 8.3333337680e-03f), -1.6666667163e-01f);
    const float sp = __builtin_fmaf(r * z, ps, r);
    const float pc = __builtin_fmaf(z, __builtin_fmaf(z, __builtin_fmaf(z, -2.7557314297e-07f, 2.4801587642e-05f), -1.3888889225e-03f), 4.1666667908e-02f);
    const float cp = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
    switch (q & 3) {
    case 0: *sn = sp; *cs = cp; break;
    case 1: *sn = cp; *cs = -sp; break;
    case 2: *sn = -sp; *cs = -cp; break;
    default: *sn = -cp; *cs = sp; break;
    }
}

// sin and cos of a phase given in units of 2 pi / 2^32 (the fine-CFO oscillator, rx_types.h: CfoSegment): the quadrant is
// the phase's top two bits (rounded), the rest a signed 30-bit number -> r in [-pi/4, pi/4) with one conversion and one
// product -- no range test, no reduction; the kernels of dm_sincos on r.  (oracle: orc_sincos_phase, the same sequence)
__device__ __forceinline__ void dm_sincos_phase(uint32_t a, float* sn, float* cs)
{
    const uint32_t q = (a + 0x20000000u) >> 30;
    const float r = float(int32_t(a - (q << 30))) * float(3.14159265358979323846 / 2147483648.0);
    const float z = r * r;
    const float ps = __builtin_fmaf(z, __builtin_fmaf(z, __builtin_fmaf(z, 2.7557314297e-06f, -1.9841270114e-04f), 8.3333337680e-03f), -1.6666667163e-01f);
    const float sp = __builtin_fmaf(r * z, ps, r);
    const float pc = __builtin_fmaf(z, __builtin_fmaf(z, __builtin_fmaf(z, -2.7557314297e-07f, 2.4801587642e-05f), -1.3888889225e-03f), 4.1666667908e-02f);
    const float cp = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
    switch (q & 3u) {
    case 0: *sn = sp; *cs = cp; break;
    case 1: *sn = cp; *cs = -sp; break;
    case 2: *sn = -sp; *cs = -cp; break;
    default: *sn = -cp; *cs = sp; break;
    }
}

} // namespace dtl
