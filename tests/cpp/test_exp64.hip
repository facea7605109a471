// dm_exp64 (gr_dtl_amd/csrc/dtl_math.h, GPU) against orc_exp (oracle/orc_math.h, CPU): the same operation sequence must
// give the same bits.  Arguments: every `stride`-th float bit pattern converted to double (the decoder's arguments are
// floats), plus doubles around the overflow / underflow / denormal edges.  Prints "OK checked=<n>" or the mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#pragma clang fp contract(off)
#include "../../gr_dtl_amd/csrc/dtl_math.h"
extern "C" {
#include "../../oracle/orc_math.h"
}

__global__ void eval(const double* __restrict__ x, double* __restrict__ y, size_t n)
{
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) y[i] = dtl::dm_exp64(x[i]);
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
    eval<<<dim3(unsigned((n + 255) / 256)), dim3(256)>>>(dx, dy, n);
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
    printf("OK checked=%zu\n", n);
    return 0;
}
