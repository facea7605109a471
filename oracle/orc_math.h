/*
 * orc_math.h -- ORACLE (test infrastructure): float expf / logf / sincosf with a FIXED, documented
 * operation sequence (IEEE single operations, fused multiply-adds written out explicitly, nothing left to
 * the compiler's contraction), <= 2 ulp from the true value.
 *
 * Why: the reference calls libm (expf/logf in constellation::calc_soft_dec, sincosf in gr_expj and
 * the oscillator); libm results differ between platforms in the last bit, and the LDPC decoder of a
 * non-converging codeword amplifies such differences into different hard decisions.  Fixing the
 * algorithm makes the soft-bit path bit-reproducible: gr_dtl_amd/csrc/dtl_math.h implements the same
 * sequences on the GPU.  Algorithms: Cody-Waite range reduction + Taylor/minimax polynomials with the
 * classic fdlibm single-precision coefficients (public domain algorithm descriptions).
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float orc_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t orc_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
#define ORC_FMA(a, b, c) __builtin_fmaf((a), (b), (c)) /* one rounding; vfmadd on x86-64-v3, v_fma_f32 on the GPU */

/* exp(x): k = rint(x*log2e); r = x - k*ln2 (hi/lo); degree-7 polynomial; two-step scaling by 2^k */
static inline float orc_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283905f) return INFINITY;
    if (x < -103.97208f) return 0.0f;
    float kf = rintf(x * 1.44269504f);
    int k = (int)kf;
    float r = ORC_FMA(-kf, 0.693145752f, x);
    r = ORC_FMA(-kf, 1.42860677e-06f, r);
    float p = 1.9841270e-4f;
    p = ORC_FMA(p, r, 1.3888889e-3f);
    p = ORC_FMA(p, r, 8.3333338e-3f);
    p = ORC_FMA(p, r, 4.1666668e-2f);
    p = ORC_FMA(p, r, 1.6666667e-1f);
    p = ORC_FMA(p, r, 0.5f);
    p = ORC_FMA(p, r, 1.0f);
    p = ORC_FMA(p, r, 1.0f);
    int k1 = k / 2, k2 = k - k1; /* |k1|,|k2| <= 75: both scale factors are normal numbers */
    float s1 = orc_bits2f((uint32_t)(k1 + 127) << 23), s2 = orc_bits2f((uint32_t)(k2 + 127) << 23);
    return (p * s1) * s2;
}

/* exp(x) in double, same construction: k = rint(x*log2e); r = x - k*ln2 (fdlibm's hi/lo split: k*ln2_hi is exact);
 * Taylor polynomial of degree 13 (|r| <= 0.347: truncation 2^-57) in Horner form, fused multiply-adds written out;
 * two-step scaling by 2^k (the first product is exact, the second rounds once -- also into the denormals).
 * < 1 ulp.  The decoder's channel likelihood ratios (awgn_bp.cc: exp of the scaled soft bit) go through this instead
 * of the C library's exp, whose last bit differs between libraries (glibc / the GPU's ocml differ at x = 40). */
#define ORC_FMA64(a, b, c) __builtin_fma((a), (b), (c))
static inline double orc_exp(double x)
{
    if (x != x) return x;
    if (x > 710.0) return (double)INFINITY;
    if (x < -746.0) return 0.0;
    double kd = rint(x * 0x1.71547652b82fep+0);
    int k = (int)kd;
    double r = ORC_FMA64(-kd, 0x1.62e42fee00000p-1, x);
    r = ORC_FMA64(-kd, 0x1.a39ef35793c76p-33, r);
    double p = 0x1.6124613a86d09p-33;
    p = ORC_FMA64(p, r, 0x1.1eed8eff8d898p-29);
    p = ORC_FMA64(p, r, 0x1.ae64567f544e4p-26);
    p = ORC_FMA64(p, r, 0x1.27e4fb7789f5cp-22);
    p = ORC_FMA64(p, r, 0x1.71de3a556c734p-19);
    p = ORC_FMA64(p, r, 0x1.a01a01a01a01ap-16);
    p = ORC_FMA64(p, r, 0x1.a01a01a01a01ap-13);
    p = ORC_FMA64(p, r, 0x1.6c16c16c16c17p-10);
    p = ORC_FMA64(p, r, 0x1.1111111111111p-7);
    p = ORC_FMA64(p, r, 0x1.5555555555555p-5);
    p = ORC_FMA64(p, r, 0x1.5555555555555p-3);
    p = ORC_FMA64(p, r, 0x1.0000000000000p-1);
    p = ORC_FMA64(p, r, 1.0);
    p = ORC_FMA64(p, r, 1.0);
    int k1 = k / 2, k2 = k - k1; /* |k1|, |k2| <= 538: both scale factors are normal numbers */
    uint64_t b1 = (uint64_t)(k1 + 1023) << 52, b2 = (uint64_t)(k2 + 1023) << 52;
    double s1, s2;
    memcpy(&s1, &b1, 8); memcpy(&s2, &b2, 8);
    return (p * s1) * s2;
}

/* log(x), log10(x) and 10^y in double, written out like the functions above: the frame equaliser's SNR / noise figures
 * (mpsk_snr_est_simple::snr() / noise(), equalizer.cc:279-287: 10 log10(.), 10^(./10)) scale every soft bit of a frame, and
 * the C library's and the GPU library's versions of these functions disagree in the last bits often enough to show.
 * log: x = m 2^k, m in [sqrt(1/2), sqrt(2)); f = m - 1, s = f / (2 + f); fdlibm's degree-14 odd polynomial in s
 * (coefficients Lg1..Lg7) and its final assembly; log10 = log * (1 / ln 10) (one more rounding: <= 2 ulp);
 * 10^y = exp(y ln10) with the product y ln10 carried in two pieces (p + e), exp(p) (1 + e).  Plain IEEE operations in the
 * order written (no contraction); the GPU executes the same sequence (dtl_math.h). */
static inline double orc_log(double x)
{
    if (x != x) return x;
    if (x < 0.0) return (double)NAN;
    if (x == 0.0) return -(double)INFINITY;
    if (x == (double)INFINITY) return x;
    int k = 0;
    if (x < 0x1p-1022) { x *= 0x1p54; k = -54; }            /* subnormal: scale into the normal range */
    uint64_t b; memcpy(&b, &x, 8);
    k += (int)(b >> 52) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL; /* m in [1, 2) */
    double m; memcpy(&m, &b, 8);
    if (m > 0x1.6a09e667f3bcdp+0) { m *= 0.5; k += 1; }      /* m in [sqrt(1/2), sqrt(2)) */
    const double f = m - 1.0, s = f / (2.0 + f), z = s * s, w = z * z, dk = (double)k;
    const double t1 = w * (0x1.999999997fa04p-2 + w * (0x1.c71c51d8e78afp-3 + w * 0x1.39a09d078c69fp-3));
    const double t2 = z * (0x1.5555555555593p-1 + w * (0x1.2492494229359p-2 + w * (0x1.7466496cb03dep-3 + w * 0x1.2f112df3e5244p-3)));
    const double R = t2 + t1, hfsq = 0.5 * f * f;
    return dk * 0x1.62e42fee00000p-1 - ((hfsq - (s * (hfsq + R) + dk * 0x1.a39ef35793c76p-33)) - f);
}
static inline double orc_log10(double x) { return orc_log(x) * 0x1.bcb7b1526e50ep-2; }
static inline double orc_pow10(double y)
{
    const double p = y * 0x1.26bb1bbb55516p+1;
    const double r = orc_exp(p);
    if (!(p > -800.0 && p < 800.0) || r == 0.0 || r == (double)INFINITY) return r; /* overflow / underflow / NaN: no correction term */
    const double e = ORC_FMA64(y, 0x1.26bb1bbb55516p+1, -p) + y * -0x1.f48ad494ea3e9p-53;
    return ORC_FMA64(r, e, r);
}

/* log(x), x >= 0: fdlibm e_logf kernel */
static inline float orc_logf(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    if (x < 1.17549435e-38f) { x = x * 33554432.0f; e = -25; } /* subnormal: scale by 2^25 */
    uint32_t ix = orc_f2bits(x);
    e += (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) | 0x3f800000u; /* m in [1,2) */
    float m = orc_bits2f(ix);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float R = z * ORC_FMA(z, ORC_FMA(z, ORC_FMA(z, 0.24279078841f, 0.28498786688f), 0.40000972152f), 0.66666662693f);
    float hfsq = 0.5f * f * f;
    float ef = (float)e;
    return ORC_FMA(ef, 6.9313812256e-01f, f - (hfsq - ORC_FMA(s, hfsq + R, ef * 9.0580006145e-06f)));
}

/* sin and cos of x, any finite x: reduction to [-pi, pi) in double, quadrant split, fdlibm kernels */
static inline void orc_sincosf(float x, float* sn, float* cs)
{
    if (!(x == x) || x == INFINITY || x == -INFINITY) { *sn = *cs = NAN; return; }
    double xd = (double)x;
    if (xd >= 3.14159265358979323846 || xd < -3.14159265358979323846) {
        double kk = floor((xd + 3.14159265358979323846) * (1.0 / 6.28318530717958647692));
        xd = xd - 6.28318530717958647692 * kk;
    }
    float xr = (float)xd;
    float qf = rintf(xr * 0.636619772f); /* 2/pi */
    int q = (int)qf;
    float r = ORC_FMA(-qf, 1.57079601e+00f, xr);  /* pi/2 hi */
    r = ORC_FMA(-qf, 3.13916473e-07f, r);         /* pi/2 lo */
    float z = r * r;
    float ps = ORC_FMA(z, ORC_FMA(z, ORC_FMA(z, 2.7557314297e-06f, -1.9841270114e-04f), 8.3333337680e-03f), -1.6666667163e-01f);
    float sp = ORC_FMA(r * z, ps, r);
    float pc = ORC_FMA(z, ORC_FMA(z, ORC_FMA(z, -2.7557314297e-07f, 2.4801587642e-05f), -1.3888889225e-03f), 4.1666667908e-02f);
    float cp = ORC_FMA(z * z, pc, ORC_FMA(-0.5f, z, 1.0f));
    switch (q & 3) {
    case 0: *sn = sp; *cs = cp; break;
    case 1: *sn = cp; *cs = -sp; break;
    case 2: *sn = -sp; *cs = -cp; break;
    default: *sn = -cp; *cs = sp; break;
    }
}

/* sin and cos of a phase in units of 2 pi / 2^32 (the fine-CFO oscillator, orc_cfo_mix_trig): quadrant = the top two bits
 * (rounded), the rest a signed 30-bit number -> r in [-pi/4, pi/4); the kernels of orc_sincosf on r */
static inline void orc_sincos_phase(uint32_t a, float* sn, float* cs)
{
    const uint32_t q = (a + 0x20000000u) >> 30;
    const float r = (float)(int32_t)(a - (q << 30)) * (float)(3.14159265358979323846 / 2147483648.0);
    float z = r * r;
    float ps = ORC_FMA(z, ORC_FMA(z, ORC_FMA(z, 2.7557314297e-06f, -1.9841270114e-04f), 8.3333337680e-03f), -1.6666667163e-01f);
    float sp = ORC_FMA(r * z, ps, r);
    float pc = ORC_FMA(z, ORC_FMA(z, ORC_FMA(z, -2.7557314297e-07f, 2.4801587642e-05f), -1.3888889225e-03f), 4.1666667908e-02f);
    float cp = ORC_FMA(z * z, pc, ORC_FMA(-0.5f, z, 1.0f));
    switch (q & 3u) {
    case 0: *sn = sp; *cs = cp; break;
    case 1: *sn = cp; *cs = -sp; break;
    case 2: *sn = -sp; *cs = -cp; break;
    default: *sn = -cp; *cs = sp; break;
    }
}

#endif
