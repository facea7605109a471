// Experiment: how accurate is v_rcp_f64 on gfx950?  max |1 - b * rcp(b)| over random mantissas (the residual is exact: one fma),
// and the number of bit mismatches against IEEE a / b of the division sequence with ONE Newton step instead of two.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#pragma clang fp contract(off)

__device__ uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

__device__ __forceinline__ double div_one_step(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

__global__ void probe(double* worst, unsigned long long* mism)
{
    uint64_t s = 0x9E3779B97F4A7C15ull ^ (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 0xD1B54A32D192ED03ull;
    double w = 0;
    unsigned long long bad = 0;
    for (int it = 0; it < 16384; it++) {
        const double b = __longlong_as_double((uint64_t(1023) << 52) | (rng(s) & 0xFFFFFFFFFFFFFull));
        const double a = __longlong_as_double((uint64_t(1023) << 52) | (rng(s) & 0xFFFFFFFFFFFFFull));
        const double e = fabs(__builtin_fma(-b, __builtin_amdgcn_rcp(b), 1.0));
        w = e > w ? e : w;
        if (__double_as_longlong(a / b) != __double_as_longlong(div_one_step(a, b))) bad++;
    }
    for (int o = 32; o; o >>= 1) { double x = __shfl_xor(w, o); w = x > w ? x : w; }
    if ((threadIdx.x & 63) == 0) { atomicMax(reinterpret_cast<unsigned long long*>(worst), (unsigned long long)__double_as_longlong(w)); }
    atomicAdd(mism, bad);
}

int main()
{
    double* d; unsigned long long* m;
    hipMalloc(&d, 8); hipMalloc(&m, 8); hipMemset(d, 0, 8); hipMemset(m, 0, 8);
    hipLaunchKernelGGL(probe, dim3(8192), dim3(256), 0, 0, d, m);
    hipDeviceSynchronize();
    double h; unsigned long long hm;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&hm, m, 8, hipMemcpyDeviceToHost);
    printf("max |1 - b rcp(b)| = %.3e = 2^%.2f over %.2e values; one-Newton-step division: %llu mismatches vs a / b\n", h, log2(h), 8192.0 * 256 * 16384, hm);
    return 0;
}
