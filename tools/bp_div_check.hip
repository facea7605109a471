// Checks on gfx950 that the shortened IEEE-division sequences of ldpc_bp.hip give the bits of `a / b` on the operand
// domains they are used on (the derivation is in the comments of div_exact / div_row_product; this is the empirical
// counterpart): random and adversarial mantissas, the extreme exponents of each domain, and every special value.
//   D1  2 / (1 + q)                     q = 0, or |q| in [2^-938, 2^938], +-inf, NaN   -> TAME form (no v_div_scale)
//   D2  product / t                     t = 2/(1+q) - 1 computed as in the kernel, product of 6 such  -> no scale, no fixup
//   D3  (1 - prd) / (1 + prd)           prd = product / t                             -> TAME form
//   D4  sum / R                         |sum| in [2^-884, 2^884] or 0 / inf / NaN, R = a D3 result (or 0 / inf / NaN) -> TAME form
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/bp_div_check.hip -o tools/bp_div_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#pragma clang fp contract(off)

__device__ __forceinline__ double div_tame(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = a * r;
    const double rem = __builtin_fma(-b, q, a);
    return __builtin_amdgcn_div_fixup(__builtin_fma(rem, r, q), b, a);
}
__device__ __forceinline__ double div_plain(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = a * r;
    const double rem = __builtin_fma(-b, q, a);
    return __builtin_fma(rem, r, q);
}
__device__ uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ bool same(double x, double y)
{
    if (x != x && y != y) return true;                      // both NaN (payloads are not observable downstream)
    return __double_as_longlong(x) == __double_as_longlong(y);
}
// a double with the given unbiased exponent; mantissa by `style`: 0 random, 1 all ones minus a little, 2 tiny, 3 zero
__device__ double make(uint64_t& s, int e, int style, bool neg)
{
    uint64_t m = rng(s) & 0xFFFFFFFFFFFFFull;
    if (style == 1) m = 0xFFFFFFFFFFFFFull - (rng(s) & 0x3FF);
    if (style == 2) m = rng(s) & 0x3FF;
    if (style == 3) m = 0;
    return __longlong_as_double((uint64_t(e + 1023) << 52) | m | (neg ? 1ull << 63 : 0));
}
__device__ double special(uint64_t& s, double dflt)
{
    switch (rng(s) % 64) {
    case 0: return 0.0;
    case 1: return -0.0;
    case 2: return __longlong_as_double(0x7FF0000000000000ull);
    case 3: return __longlong_as_double(0xFFF0000000000000ull);
    case 4: return __longlong_as_double(0x7FF8000000000000ull);
    default: return dflt;
    }
}

__global__ void check(unsigned long long* mism, unsigned long long* total, double* ex)
{
    uint64_t s = 0x9E3779B97F4A7C15ull ^ (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 0xD1B54A32D192ED03ull;
    unsigned long long bad[4] = { 0, 0, 0, 0 };
    auto note = [&](int k, double a, double b, double want, double got) {
        if (!same(want, got)) { if (bad[k]++ == 0) { ex[k * 4 + 0] = a; ex[k * 4 + 1] = b; ex[k * 4 + 2] = want; ex[k * 4 + 3] = got; } }
    };
    for (int it = 0; it < 2048; it++) {
        // ---- a row of 6 incoming ratios q, as the check update sees them
        double tt[6], product = 1.0;
        for (int j = 0; j < 6; j++) {
            const int kind = int(rng(s) % 8);
            int e = int(rng(s) % 1877) - 938;                               // |q| in [2^-938, 2^938]
            if (kind == 0) e = int(rng(s) % 7) - 3;                         // q near 1: t near 0
            if (kind == 1) e = 938 - int(rng(s) % 4);
            if (kind == 2) e = -938 + int(rng(s) % 4);
            double q = make(s, e, int(rng(s) % 4), (rng(s) % 16) == 0);     // sometimes negative (rounding can do that)
            if (kind == 3) q = -1.0 + (double(int(rng(s) % 5) - 2)) * 0x1p-53;  // 1 + q = 0 or a few ulps
            if (kind == 4) q = 1.0;                                         // t == 0: 0 / 0 further down
            q = special(s, q);
            const double d = 1 + q;
            const double w = 2.0 / d, g = div_tame(2.0, d);
            note(0, 2.0, d, w, g);
            tt[j] = w - 1;
            product = product * tt[j];
        }
        for (int j = 0; j < 6; j++) {
            const double prd = product / tt[j], prd2 = div_plain(product, tt[j]);
            // D2's quotient has ONE consumer, R = (1 - prd) / (1 + prd), and is compared through it: a zero quotient
            // comes out as +0 where IEEE gives +-0 (1 -+ 0 = 1 either way), and an infinite row product (a factor
            // 2/0 - 1) over a finite factor comes out as NaN where IEEE gives inf (R = -inf/inf = NaN either way).
            const double n = 1 - prd, d = 1 + prd;
            const double R = n / d, R2 = div_tame(n, d);
            note(2, n, d, R, R2);
            note(1, product, tt[j], R, div_tame(1 - prd2, 1 + prd2));
            // ---- a column product divided by this ratio
            double sum = make(s, int(rng(s) % 1769) - 884, int(rng(s) % 4), false);
            if ((rng(s) % 8) == 0) sum = sum * R;                           // sum is a multiple of R in the decoder
            sum = special(s, sum);
            const double Rs = special(s, R);
            note(3, sum, Rs, sum / Rs, div_tame(sum, Rs));
        }
    }
    for (int k = 0; k < 4; k++) atomicAdd(&mism[k], bad[k]);
    atomicAdd(total, 2048ull * 6);
}

int main()
{
    unsigned long long *d, h[5];
    double *ex, hex_[16];
    hipMalloc(&d, 40); hipMalloc(&ex, 128);
    hipMemset(d, 0, 40); hipMemset(ex, 0, 128);
    hipLaunchKernelGGL(check, dim3(16384), dim3(256), 0, 0, d, d + 4, ex);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(h, d, 40, hipMemcpyDeviceToHost); hipMemcpy(hex_, ex, 128, hipMemcpyDeviceToHost);
    const char* name[4] = { "D1 2/(1+q)          [tame form]", "D2 product/t        [no scale, no fixup; compared through R]",
                            "D3 (1-prd)/(1+prd)  [tame form]", "D4 sum/R            [tame form]" };
    int rc = 0;
    for (int k = 0; k < 4; k++) {
        printf("%s: %llu mismatches in %llu operand pairs", name[k], h[k], h[4]);
        if (h[k]) { printf("  (first: a=%a b=%a ieee=%a got=%a)", hex_[4 * k], hex_[4 * k + 1], hex_[4 * k + 2], hex_[4 * k + 3]); rc = 1; }
        printf("\n");
    }
    return rc;
}
