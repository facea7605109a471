// Experiment: is the unscaled Newton/Markstein f64 division bit-identical to IEEE a/b on gfx950 for
// operands in a "safe" exponent range?  Prints the number of mismatches over random + adversarial inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#pragma clang fp contract(off)

__device__ __forceinline__ double fast_div(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);            // v_rcp_f64 (approximate)
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

__device__ uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

__global__ void check(unsigned long long* mism, unsigned long long* total, int mode, double* ex)
{
    uint64_t s = 0x9E3779B97F4A7C15ull ^ (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 0xD1B54A32D192ED03ull ^ uint64_t(mode) << 40;
    unsigned long long bad = 0;
    for (int it = 0; it < 4096; it++) {
        uint64_t ua = rng(s), ub = rng(s);
        // exponent in [-300, 300]; mantissas random or adversarial (all ones / near powers of two)
        int ea = int(rng(s) % 601) - 300, eb = int(rng(s) % 601) - 300;
        if (mode == 1) { ea = int(rng(s) % 41) - 20; eb = int(rng(s) % 41) - 20; }
        uint64_t ma = ua & 0xFFFFFFFFFFFFFull, mb = ub & 0xFFFFFFFFFFFFFull;
        if (mode == 2) { mb = 0xFFFFFFFFFFFFFull - (rng(s) & 0xFF); ma = rng(s) & 0xFFF; }
        if (mode == 3) { mb = rng(s) & 0x3FF; ma = 0xFFFFFFFFFFFFFull - (rng(s) & 0x3FF); }
        double a = __longlong_as_double((uint64_t(ea + 1023) << 52) | ma | ((ua >> 63) << 63));
        double b = __longlong_as_double((uint64_t(eb + 1023) << 52) | mb | ((ub >> 63) << 63));
        if (mode == 4) { // BP-like operands: (1-p)/(1+p), 2/(1+q)
            double p = __longlong_as_double((uint64_t(1023 - 1 - (rng(s) % 50)) << 52) | ma); // (0,1)
            if (ua >> 63) p = -p;
            a = 1 - p; b = 1 + p;
        }
        double q0 = a / b, q1 = fast_div(a, b);
        if (__double_as_longlong(q0) != __double_as_longlong(q1)) { bad++; if (bad == 1 && threadIdx.x == 0) { ex[0] = a; ex[1] = b; ex[2] = q0; ex[3] = q1; } }
    }
    atomicAdd(mism, bad);
    atomicAdd(total, 4096ull);
}

int main()
{
    unsigned long long *d, h[2];
    double* ex; double hex_[4];
    hipMalloc(&d, 16); hipMalloc(&ex, 32);
    for (int mode = 0; mode < 5; mode++) {
        hipMemset(d, 0, 16); hipMemset(ex, 0, 32);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, d + 1, mode, ex);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); hipMemcpy(hex_, ex, 32, hipMemcpyDeviceToHost);
        printf("mode %d: mismatches %llu of %llu  (example a=%a b=%a ieee=%a fast=%a)\n", mode, h[0], h[1], hex_[0], hex_[1], hex_[2], hex_[3]);
    }
    return 0;
}
