// Exhaustive checks (all 2^32 float bit patterns) that the kernel-side forms of dtl_math.h return the bits of
// the plain forms the test oracle executes on the CPU (oracle/orc_math.h):
//   dm_exp: branch-free, scaling tail `ldexpf(p, k)` (one v_ldexp_f32)  vs  early returns + two-step
//           power-of-two scaling `(p * 2^(k/2)) * 2^(k - k/2)`;
//   dm_log: branch-free, division without v_div_scale / v_div_fixup     vs  early returns + IEEE division;
// plus the demapper's exact-reciprocal quotient on random operand pairs.  Prints mismatch counts (expected 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#pragma clang fp contract(off)
#include "../gr_dtl_amd/csrc/dtl_math.h"

__device__ __forceinline__ float exp_two_step(float x)
{
    if (x != x) return x;
    if (x > 88.72283905f) return __int_as_float(0x7f800000);
    if (x < -103.97208f) return 0.0f;
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
    const int k1 = k / 2, k2 = k - k1;
    const float s1 = __int_as_float((k1 + 127) << 23), s2 = __int_as_float((k2 + 127) << 23);
    return (p * s1) * s2;
}

// the early-return form with a plain IEEE division: what oracle/orc_math.h executes on the CPU
__device__ float log_plain(float x)
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
    const float R = z * __builtin_fmaf(z, __builtin_fmaf(z, __builtin_fmaf(z, 0.24279078841f, 0.28498786688f), 0.40000972152f), 0.66666662693f);
    const float hfsq = 0.5f * f * f;
    const float ef = float(e);
    return __builtin_fmaf(ef, 6.9313812256e-01f, f - (hfsq - __builtin_fmaf(s, hfsq + R, ef * 9.0580006145e-06f)));
}

__global__ void check_log(unsigned long long* mism, unsigned* example)
{
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    unsigned long long bad = 0;
    for (uint64_t u = tid; u < (1ull << 32); u += uint64_t(gridDim.x) * blockDim.x) {
        const float x = __int_as_float(int(unsigned(u)));
        const float a = log_plain(x), b = dtl::dm_log(x);
        const bool same = (__float_as_int(a) == __float_as_int(b)) || (a != a && b != b);
        if (!same) { bad++; example[0] = unsigned(u); }
    }
    if (bad) atomicAdd(mism, bad);
}

__global__ void check(unsigned long long* mism, unsigned* example)
{
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    unsigned long long bad = 0;
    for (uint64_t u = tid; u < (1ull << 32); u += uint64_t(gridDim.x) * blockDim.x) {
        const float x = __int_as_float(int(unsigned(u)));
        const float a = exp_two_step(x), b = dtl::dm_exp(x);
        const bool same = (__float_as_int(a) == __float_as_int(b)) || (a != a && b != b);
        if (!same) { bad++; example[0] = unsigned(u); }
    }
    if (bad) atomicAdd(mism, bad);
}

// float(double(a) * (-1.0 / double(b))) == -a / b (IEEE single) for the demapper's operands
__global__ void check_div(unsigned long long* mism, unsigned long long* total, unsigned* example)
{
    uint64_t st = 0x9E3779B97F4A7C15ull ^ (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 0xD1B54A32D192ED03ull;
    auto rng = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    unsigned long long bad = 0, cnt = 0;
    for (int ib = 0; ib < 64; ib++) {
        // divisor: any positive normal float with exponent in [-40, 40] (sigma = sqrt(noise power))
        const unsigned eb = 127 - 40 + unsigned(rng() % 81);
        unsigned mb = unsigned(rng()) & 0x7fffffu;
        if ((ib & 7) == 7) mb = 0x7fffffu - (unsigned(rng()) & 0xff); // all-ones significands (Markstein's hard case)
        const float b = __int_as_float(int((eb << 23) | mb));
        const double ninv = -1.0 / double(b);
        for (int ia = 0; ia < 1024; ia++) {
            const unsigned ea = 127 - 60 + unsigned(rng() % 81);
            unsigned ma = unsigned(rng()) & 0x7fffffu;
            if ((ia & 15) == 15) ma = unsigned(rng()) & 0xffu;
            const float a = __int_as_float(int((ea << 23) | ma));
            const float q0 = -a / b, q1 = float(double(a) * ninv);
            const float q2 = dtl::dm_neg_div_by(a, b, float(1.0 / double(b)));
            if (!(fabsf(q0) < 1.17549435e-38f) && __float_as_int(q0) != __float_as_int(q2)) { bad++; example[0] = unsigned(__float_as_int(a)); example[1] = unsigned(__float_as_int(b)); }
            cnt++;
            const bool tiny = fabsf(q0) < 1.17549435e-38f;
            // subnormal quotients may differ (double rounding) but exp() of both is exactly 1
            const bool same = tiny ? (dtl::dm_exp(q0) == dtl::dm_exp(q1)) : (__float_as_int(q0) == __float_as_int(q1));
            if (!same) { bad++; example[0] = unsigned(__float_as_int(a)); example[1] = unsigned(__float_as_int(b)); }
        }
    }
    atomicAdd(mism, bad);
    atomicAdd(total, cnt);
}

// dm_div_by_count(a, n, RN(1/n)) vs a / n in double: n = 1 .. 16384, a = differences of values near 1 (the SNR
// estimator's operands), small integers and their products, and arbitrary finite doubles
__global__ void check_div_count(unsigned long long* mism, unsigned long long* total, unsigned long long* example)
{
    uint64_t st = 0xD1B54A32D192ED03ull ^ (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    auto rng = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    unsigned long long bad = 0, cnt = 0;
    for (int it = 0; it < 16384; it++) {
        const double n = double(1 + (rng() % 16384));
        const double r = 1.0 / n;
        double a;
        switch (it & 3) {
        case 0: a = double(__int_as_float(int(0x3f000000u + (rng() & 0xffffffu)))) - double(__int_as_float(int(0x3f000000u + (rng() & 0xffffffu)))); break;
        case 1: a = double(rng() % 16384) * double(1 + rng() % 16384); break;
        case 2: a = __longlong_as_double((long long)((rng() & 0x800fffffffffffffull) | ((uint64_t)(1023 - 60 + rng() % 120) << 52))); break;
        default: a = __longlong_as_double((long long)rng()); break;
        }
        const double got = dtl::dm_div_by_count(a, n, r), want = a / n;
        cnt++;
        const bool same = __double_as_longlong(got) == __double_as_longlong(want) || (got != got && want != want);
        if (!same) { bad++; example[0] = (unsigned long long)__double_as_longlong(a); example[1] = (unsigned long long)n; }
    }
    atomicAdd(mism, bad);
    atomicAdd(total, cnt);
}

int main()
{
    unsigned long long *d, h = 0;
    unsigned *ex, hex_ = 0;
    (void)hipMalloc(&d, 8); (void)hipMalloc(&ex, 4);
    (void)hipMemset(d, 0, 8); (void)hipMemset(ex, 0, 4);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, ex);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hex_, ex, 4, hipMemcpyDeviceToHost);
    printf("dm_exp ldexp tail vs two-step scaling: %llu mismatches over 2^32 inputs (last example bits 0x%08x)\n", h, hex_);
    (void)hipMemset(d, 0, 8); (void)hipMemset(ex, 0, 4);
    hipLaunchKernelGGL(check_log, dim3(4096), dim3(256), 0, 0, d, ex);
    (void)hipDeviceSynchronize();
    unsigned long long hl = 0;
    (void)hipMemcpy(&hl, d, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hex_, ex, 4, hipMemcpyDeviceToHost);
    printf("dm_log (branch-free, in-range division) vs early-return form with IEEE division: %llu mismatches over 2^32 inputs (last example bits 0x%08x)\n", hl, hex_);
    unsigned long long *d2, h2[2] = { 0, 0 };
    unsigned *ex2, hex2[2] = { 0, 0 };
    (void)hipMalloc(&d2, 16); (void)hipMalloc(&ex2, 8);
    (void)hipMemset(d2, 0, 16); (void)hipMemset(ex2, 0, 8);
    hipLaunchKernelGGL(check_div, dim3(8192), dim3(256), 0, 0, d2, d2 + 1, ex2);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h2, d2, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(hex2, ex2, 8, hipMemcpyDeviceToHost);
    printf("float(double(a) * (-1/b)) and the two-correction f32 form vs -a/b: %llu mismatches over %llu operand pairs (example a=0x%08x b=0x%08x)\n",
           h2[0], h2[1], hex2[0], hex2[1]);
    unsigned long long *d3 = nullptr, h3[4] = { 0, 0, 0, 0 };
    (void)hipMalloc(&d3, 32); (void)hipMemset(d3, 0, 32);
    hipLaunchKernelGGL(check_div_count, dim3(8192), dim3(256), 0, 0, d3, d3 + 1, d3 + 2);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h3, d3, 32, hipMemcpyDeviceToHost);
    printf("dm_div_by_count (table reciprocal, two corrections) vs a / n in double: %llu mismatches over %llu operand pairs (example a=0x%016llx n=%llu)\n",
           h3[0], h3[1], h3[2], h3[3]);
    return (h != 0) || (h2[0] != 0) || (hl != 0) || (h3[0] != 0);
}
