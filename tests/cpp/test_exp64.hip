// dm_exp64 (gr_dtl_amd/csrc/dtl_math.h, GPU) against orc_exp (oracle/orc_math.h, CPU): the same operation sequence must
// give the same bits.  Arguments: every `stride`-th float bit pattern converted to double (the decoder's arguments are
// floats), plus doubles around the overflow / underflow / denormal edges.  Prints "OK checked=<n>" or the mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#pragma clang fp contract(off)
#include "../../gr_dtl_amd/csrc/dtl_math.h"
extern "C" {
#include "../../oracle/orc_math.h"
}

__global__ void eval(const double* __restrict__ x, double* __restrict__ y, size_t n, int which)
{
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) y[i] = which == 0 ? dtl::dm_exp64(x[i]) : which == 1 ? dtl::dm_log10_64(x[i]) : dtl::dm_pow10_64(x[i]);
}

// log10 / 10^y (the frame's SNR and noise figures): doubles of every magnitude, the special values, the subnormal range
static int check_fn(int which, const char* name, const std::vector<double>& xs)
{
    const size_t n = xs.size();
    double *dx = nullptr, *dy = nullptr;
    if (hipMalloc(&dx, n * 8) != hipSuccess || hipMalloc(&dy, n * 8) != hipSuccess) { printf("FAIL hipMalloc\n"); return 2; }
    hipMemcpy(dx, xs.data(), n * 8, hipMemcpyHostToDevice);
    eval<<<dim3(unsigned((n + 255) / 256)), dim3(256)>>>(dx, dy, n, which);
    std::vector<double> ys(n);
    if (hipMemcpy(ys.data(), dy, n * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL copy\n"); return 2; }
    hipFree(dx); hipFree(dy);
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) {
        const double want = which == 1 ? orc_log10(xs[i]) : orc_pow10(xs[i]);
        if (memcmp(&want, &ys[i], 8) != 0 && !(want != want && ys[i] != ys[i])) {
            if (bad++ < 8) printf("%s x=%a gpu=%a cpu=%a\n", name, xs[i], ys[i], want);
        }
    }
    if (bad) { printf("FAIL %s mismatches=%zu of %zu\n", name, bad, n); return 1; }
    return 0;
}

int main(int argc, char** argv)
{
    const uint32_t stride = argc > 1 ? uint32_t(atoi(argv[1])) : 97u;
    std::vector<double> xs;
    for (uint64_t u = 0; u <= 0xffffffffull; u += stride) {
        float f; const uint32_t b = uint32_t(u); memcpy(&f, &b, 4);
        xs.push_back(double(f));
    }
    for (double c : { 709.0, 709.78, 710.0, -708.0, -745.13, -746.0, 40.0, -40.0, 0.0 })
        for (int j = -2000; j <= 2000; j++) xs.push_back(c + j * 1e-3 + j * 1e-13);
    const size_t n = xs.size();
    double *dx = nullptr, *dy = nullptr;
    if (hipMalloc(&dx, n * 8) != hipSuccess || hipMalloc(&dy, n * 8) != hipSuccess) { printf("FAIL hipMalloc\n"); return 2; }
    hipMemcpy(dx, xs.data(), n * 8, hipMemcpyHostToDevice);
    eval<<<dim3(unsigned((n + 255) / 256)), dim3(256)>>>(dx, dy, n, 0);
    std::vector<double> ys(n);
    if (hipMemcpy(ys.data(), dy, n * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL copy\n"); return 2; }
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) {
        const double want = orc_exp(xs[i]);
        if (memcmp(&want, &ys[i], 8) != 0 && !(want != want && ys[i] != ys[i])) {
            if (bad++ < 8) printf("x=%a gpu=%a cpu=%a\n", xs[i], ys[i], want);
        }
    }
    if (bad) { printf("FAIL mismatches=%zu of %zu\n", bad, n); return 1; }
    {
        std::vector<double> a, b;
        uint64_t st = 0x9e3779b97f4a7c15ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
        for (int i = 0; i < 6000000; i++) {
            uint64_t bits = rnd() & 0x7fffffffffffffffull;           // any non-negative double, NaN and inf included
            double v; memcpy(&v, &bits, 8);
            a.push_back(v);
            b.push_back((double(rnd() >> 11) * 0x1p-53) * 700.0 - 350.0);
        }
        for (int e = -1080; e <= 1030; e++) { a.push_back(ldexp(1.0, e)); a.push_back(ldexp(1.4142135623730951, e)); a.push_back(ldexp(1.4142135623730949, e)); }
        for (double v : { 0.0, -0.0, -1.0, 1.0, 10.0 }) a.push_back(v);
        for (double v : { 0.0, -0.0, 1.0, -1.0, 308.0, 308.3, 309.0, -307.0, -320.0, -324.0, -330.0, 1e300, -1e300 }) b.push_back(v);
        const double inf = 1.0 / 0.0;
        b.push_back(inf); b.push_back(-inf); b.push_back(inf - inf);
        if (int rc = check_fn(1, "log10", a)) return rc;
        if (int rc = check_fn(2, "pow10", b)) return rc;
    }
    printf("OK checked=%zu\n", n);
    return 0;
}
