// Issue cost of the single-precision division forms on gfx950: the compiler's IEEE sequence (v_div_scale x2, v_rcp,
// FMAs, v_div_fmas, v_div_fixup), the same refinement without scaling / fix-up, and the refinement with a known
// reciprocal (dm_neg_div_by's form); plus a plain FMA loop as the yardstick.  Prints ns and SIMD cycles per operation.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)

constexpr int kIter = 2048, kChains = 8;

template <int MODE>
__global__ __launch_bounds__(256) void bench(float* out, float seed)
{
    float a[kChains], acc[kChains];
    for (int c = 0; c < kChains; c++) { a[c] = seed + 0.001f * float(threadIdx.x + c); acc[c] = 0.0f; }
    const float b0 = 1.0f + seed;
    for (int it = 0; it < kIter; it++) {
#pragma unroll
        for (int c = 0; c < kChains; c++) {
            const float n = a[c], d = b0 + acc[c] * 1e-9f;
            float q;
            if (MODE == 0) {
                q = n / d;
            } else if (MODE == 1) {
                const float y0 = __builtin_amdgcn_rcpf(d);
                const float e0 = __builtin_fmaf(-d, y0, 1.0f);
                const float y1 = __builtin_fmaf(e0, y0, y0);
                const float q0 = n * y1;
                const float r0 = __builtin_fmaf(-d, q0, n);
                const float q1 = __builtin_fmaf(r0, y1, q0);
                const float r1 = __builtin_fmaf(-d, q1, n);
                q = __builtin_fmaf(r1, y1, q1);
            } else if (MODE == 2) {
                const float r = seed; // a "known reciprocal"
                float q0 = n * r;
                float e = __builtin_fmaf(-d, q0, n);
                q0 = __builtin_fmaf(e, r, q0);
                e = __builtin_fmaf(-d, q0, n);
                q = __builtin_fmaf(e, r, q0);
            } else {
                q = __builtin_fmaf(n, d, 0.5f);
            }
            acc[c] += q;
            a[c] = a[c] + 1e-6f;
        }
    }
    float s = 0;
    for (int c = 0; c < kChains; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * 8;
    hipLaunchKernelGGL(bench<MODE>, dim3(grid), dim3(256), 0, 0, d, 0.37f);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(bench<MODE>, dim3(grid), dim3(256), 0, 0, d, 0.37f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double wave_ops = double(grid) * 4 * kIter * kChains;            // operations per wave-instruction stream
    const double per_simd = wave_ops / 1024.0;                              // per SIMD
    printf("%-46s %8.3f ms  %6.2f ns per wave-op per SIMD  (= %5.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4);
}

int main()
{
    float* d = nullptr;
    (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<3>("fma + 2 adds (yardstick)", d);
    run<0>("IEEE a / b (+ 2 adds, 1 fma for the divisor)", d);
    run<1>("rcp + refinement, no scaling / fix-up", d);
    run<2>("refinement with a known reciprocal", d);
    (void)hipFree(d);
    return 0;
}
