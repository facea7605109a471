// rx_kernels.hip -- HIP kernels (gfx950) of the adaptive-OFDM receive path, K1..K9/K11.
//
// Layout: the IQ stream stays where the caller put it (complex64, device).  One workgroup owns
// one OFDM frame; its (2+h) preamble/header symbols (phase A) or its L payload symbols
// (phase B) are pulled through LDS once: coalesced 8-byte loads (lane == sample), in-LDS
// radix-2 FFT, then lane == subcarrier for channel estimation / equalisation / demapping.
// No MFMA anywhere: nothing here is a dense contraction.
//
// Arithmetic follows the CPU reference chain (see each kernel); complex division is Smith's
// algorithm and complex multiplication the 4-multiply form, both evaluated in float without
// FMA contraction, i.e. exactly what libgcc's __divsc3/__mulsc3 give the reference's
// std::complex<float> code on x86-64.
#include "rx_kernels.h"

#include "dtl_ablate.h"   // timing experiments: every `ablated(k)` below is false in the shipped library
#include "dtl_math.h"

#include <algorithm>
#include <mutex>
#include <cstdlib>

#pragma clang fp contract(off)

namespace dtl {

namespace {

__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b)
{
    return { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
}
// twiddle product of the FFT butterflies: one fused multiply-add per component (the oracle's FFT does the same)
__device__ __forceinline__ cf32 cmul_tw(cf32 a, cf32 w)
{
    return { __builtin_fmaf(a.re, w.re, -(a.im * w.im)), __builtin_fmaf(a.re, w.im, a.im * w.re) };
}
__device__ __forceinline__ cf32 cadd(cf32 a, cf32 b) { return { a.re + b.re, a.im + b.im }; }
__device__ __forceinline__ cf32 csub(cf32 a, cf32 b) { return { a.re - b.re, a.im - b.im }; }
__device__ __forceinline__ cf32 cscale(float s, cf32 a) { return { s * a.re, s * a.im }; }
__device__ __forceinline__ cf32 cconj(cf32 a) { return { a.re, -a.im }; }
__device__ __forceinline__ bool cnonzero(cf32 a) { return a.re != 0.0f || a.im != 0.0f; }

// (a+ib)/(c+id), Smith's algorithm in float (libgcc __divsc3 for finite operands).  Its two cases
//   |c| <  |d|:  r = c/d, den = c r + d, ((a r + b) / den, (b r - a) / den)
//   |c| >= |d|:  r = d/c, den = d r + c, ((b r + a) / den, (b - a r) / den)
// are one sequence of operations on swapped operands (only the last subtraction changes direction), so the lanes of
// a wave -- which split between the cases at random -- share the three divisions instead of executing six.
__device__ __forceinline__ cf32 cdiv(cf32 x, cf32 y)
{
    const bool sw = fabsf(y.re) < fabsf(y.im);
    const float p = sw ? y.re : y.im, qd = sw ? y.im : y.re;
    const float m = sw ? x.re : x.im, n = sw ? x.im : x.re;
    const float r = p / qd;
    const float den = (p * r) + qd;
    const float re_num = (m * r) + n;
    const float t = n * r;
    const float im_num = sw ? (t - m) : (m - t);
    return { re_num / den, im_num / den };
}

__device__ __forceinline__ cf32 expj(float phase)
{
    float s, c;
    dm_sincos(phase, &s, &c);
    return { c, s };
}

constexpr double kPi = 3.14159265358979323846;

// Oscillator phase at stream index i: frequency_modulator_fc(-2/N) integrating the held angle.
// The segments a frame can touch are loaded once per workgroup (uniform values): the one active at the frame's
// trigger and its successor; an index beyond the successor's end takes the general walk.
struct SegWindow {
    CfoSegment a, b;
    int64_t c_start;   // start of the segment after b (or the largest index)
    int seg;           // index of a (-1: before the first trigger)
    bool has_a, has_b;
};

__device__ __forceinline__ SegWindow seg_window(const CfoSegment* __restrict__ segs, int n_segs, int seg)
{
    SegWindow w{};
    w.seg = seg;
    w.has_a = seg >= 0 && seg < n_segs;
    if (w.has_a) w.a = segs[seg];
    w.has_b = seg + 1 >= 0 && seg + 1 < n_segs;
    if (w.has_b) w.b = segs[seg + 1];
    w.c_start = (seg + 2 >= 0 && seg + 2 < n_segs) ? segs[seg + 2].start : 0x7fffffffffffffffLL;
    return w;
}

__device__ __forceinline__ uint32_t osc_phase(const CfoSegment* __restrict__ segs, int n_segs, const SegWindow& w,
                                              int64_t i, float /*sens*/)
{
    // phase0 + inc (i - start + 1) in units of 2 pi / 2^32, wrapping by itself (CfoSegment, rx_types.h); before the first
    // trigger the held angle is 0 and the phase stays 0
    uint32_t a = 0u;
    if (i >= w.c_start) {
        int seg = w.seg + 2;
        while (seg + 1 < n_segs && segs[seg + 1].start <= i) seg++;
        const CfoSegment s = segs[seg];
        a = s.phase0 + s.inc * uint32_t(i - s.start + 1);
    } else if (w.has_b && w.b.start <= i) {
        a = w.b.phase0 + w.b.inc * uint32_t(i - w.b.start + 1);
    } else if (w.has_a && w.a.start <= i) {
        a = w.a.phase0 + w.a.inc * uint32_t(i - w.a.start + 1);
    }
    return a;
}
// the phases of the four samples i, i + stride, i + 2 stride, i + 3 stride (a lane's samples of one OFDM symbol): when they
// lie inside ONE of the window's two segments -- nearly always -- one product for the first and three additions of
// inc x stride (the same numbers mod 2^32 as four evaluations); otherwise sample by sample
__device__ __forceinline__ void osc_phase4(const CfoSegment* __restrict__ segs, int n_segs, const SegWindow& w, int64_t i,
                                           int stride, float sens, uint32_t (&a)[4])
{
    const int64_t last = i + 3 * int64_t(stride);
    const bool in_b = w.has_b && w.b.start <= i && last < w.c_start;
    const bool in_a = w.has_a && w.a.start <= i && last < (w.has_b ? w.b.start : w.c_start);
    if (in_a || in_b) {
        const uint32_t p0 = in_b ? w.b.phase0 : w.a.phase0, inc = in_b ? w.b.inc : w.a.inc;
        const int64_t st = in_b ? w.b.start : w.a.start;
        const uint32_t step = inc * uint32_t(stride);
        a[0] = p0 + inc * uint32_t(i - st + 1);
        a[1] = a[0] + step; a[2] = a[1] + step; a[3] = a[2] + step;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = osc_phase(segs, n_segs, w, i + int64_t(j) * stride, sens);
    }
}
__device__ __forceinline__ cf32 expj_phase(uint32_t a)
{
    float s, c;
    dm_sincos_phase(a, &s, &c);
    return { c, s };
}

// Load up to four consecutive OFDM symbols (sym0 .. sym0+n_sym-1) of the frame cut at `trigger` (delay N+cp,
// CP stripped, fine-CFO derotated) into the rows buf + u*N in bit-reversed order, then radix-2 DIT FFTs in
// place.  blockDim.x == N.  Two butterfly stages are executed per LDS round trip: a thread owns the four
// elements {i, i+h, i+2h, i+3h} of one symbol, runs the two span-h butterflies of stage s and the two span-2h
// butterflies of stage s+1 on them in registers (the same operations, in the same order per butterfly, as two
// separate passes -- bit-identical), so the N/4 groups of four symbols keep all N lanes busy with half the
// LDS traffic and half the barriers.  An odd last stage runs as a plain pass.  tw: the N per-pass twiddles in LDS;
// rows are stored LDS-swizzled (fft_lds_swizzle, rx_types.h): readers index them through it.
template <int ABL = 0> // timing experiments (dtl_ablate.h): 1 / 42 no derotation, 2 / 43 no butterflies
__device__ __forceinline__ void load_and_fft4(const RxTables& t, const cf32* __restrict__ iq, int64_t n_samples,
                                              int64_t trigger, int sym0, int n_sym, const CfoSegment* segs,
                                              int n_segs, const SegWindow& seg, cf32* buf, const cf32* tw)
{
    const int N = t.fft_len, P = N + t.cp_len;
    const int tid = threadIdx.x;
    const float sens = float(-2.0 / N);
    const int logN = __ffs(N) - 1;
    const int u = tid >> (logN - 2), q = tid & ((N >> 2) - 1); // symbol of the batch, group of four inside it
    cf32* x = buf + u * N;
    const bool active = u < n_sym;
    // Up to N = 256 the N/4 lanes that transform one symbol sit in one wave.  Then the same lanes also load that
    // symbol (four samples each), and between the load and the last butterfly pass nothing is shared across
    // waves: those steps are ordered by the wave's own program order, and the workgroup meets only once, at the end.
    const bool wave_local = (N >> 2) <= 64;
    auto sync_stage = [&]() {
        if (wave_local) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };
    // mixed[i] = r[i - (N+cp)] * osc[i]; sample s of symbol sym sits at stream index i = trigger + sym*P + cp + s.
    // The loads carry no branch (address clamped into the stream, value zeroed afterwards): all of a lane's samples
    // and row positions are requested before the first one is waited for -- with a bounds test around each load the
    // compiler serialised them, one exposed memory latency per sample.
    if (wave_local) {
        if (active) {
            const int64_t i0 = trigger + int64_t(sym0 + u) * P + t.cp_len;
            cf32 v[4];
            int br[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int sidx = q + j * (N >> 2);
                const int64_t src = i0 + sidx - P;
                const bool inside = src >= 0 && src < n_samples;
                const cf32 val = iq[inside ? src : 0];
                br[j] = t.bitrev[sidx];               // bit-reversed and swizzled
                v[j] = cf32{ inside ? val.re : 0.0f, inside ? val.im : 0.0f };
            }
            uint32_t ph[4] = { 0u, 0u, 0u, 0u };
            if constexpr (!(ABL == 1 || ABL == 42)) osc_phase4(segs, n_segs, seg, i0 + q, N >> 2, sens, ph);
#pragma unroll
            for (int j = 0; j < 4; j++) x[br[j]] = (ABL == 1 || ABL == 42) ? v[j] : cmul(v[j], expj_phase(ph[j]));
        }
    } else {
        const int br = t.bitrev[tid];                 // bit-reversed and swizzled
        cf32 v[4];
#pragma unroll
        for (int uu = 0; uu < 4; uu++) {
            const int64_t src = trigger + int64_t(sym0 + uu) * P + t.cp_len + tid - P;
            const bool inside = uu < n_sym && src >= 0 && src < n_samples;
            const cf32 val = iq[inside ? src : 0];
            v[uu] = cf32{ inside ? val.re : 0.0f, inside ? val.im : 0.0f };
        }
        uint32_t ph[4] = { 0u, 0u, 0u, 0u };           // (a thread's four samples are one symbol period apart)
        if constexpr (!(ABL == 1 || ABL == 42)) osc_phase4(segs, n_segs, seg, trigger + int64_t(sym0) * P + t.cp_len + tid, P, sens, ph);
#pragma unroll
        for (int uu = 0; uu < 4; uu++)
            if (uu < n_sym) buf[uu * N + br] = (ABL == 1 || ABL == 42) ? v[uu] : cmul(v[uu], expj_phase(ph[uu]));
    }
    sync_stage();
    int st = 0;
    if (ABL == 2 || ABL == 43) st = logN;
    for (; st + 1 < logN; st += 2) {
        if (active) {
            const int h = 1 << st;
            const int k = q & (h - 1);
            // elements i0, i0 + h, i0 + 2h, i0 + 3h: bits st, st+1 of i0 are clear, so the swizzle of the sum is the
            // XOR of the swizzles (the three offsets are wave-uniform)
            const int i0 = fft_lds_swizzle(((q >> st) << (st + 2)) + k);
            const int i1 = i0 ^ fft_lds_swizzle(h), i2 = i0 ^ fft_lds_swizzle(2 * h), i3 = i0 ^ fft_lds_swizzle(3 * h);
            const cf32* twp = tw + (h - 1);
            // stage st: (i0,i1) and (i2,i3), twiddle W^(k N / 2h)
            const cf32 w0 = twp[k];
            cf32 a0 = x[i0], a1 = x[i1], a2 = x[i2], a3 = x[i3];
            cf32 v = cmul_tw(a1, w0);
            a1 = csub(a0, v); a0 = cadd(a0, v);
            v = cmul_tw(a3, w0);
            a3 = csub(a2, v); a2 = cadd(a2, v);
            // stage st+1: (i0,i2) with W^(k N / 4h), (i1,i3) with W^((k+h) N / 4h)
            const cf32 w1 = twp[h + k], w2 = twp[2 * h + k];
            v = cmul_tw(a2, w1);
            a2 = csub(a0, v); a0 = cadd(a0, v);
            v = cmul_tw(a3, w2);
            a3 = csub(a1, v); a1 = cadd(a1, v);
            x[i0] = a0; x[i1] = a1; x[i2] = a2; x[i3] = a3;
        }
        sync_stage();
    }
    if (st < logN) { // log2(N) odd: one plain stage, N/2 butterflies per symbol = two per thread
        if (active) {
            const int half = 1 << st;
            const cf32* twn = tw + (half - 1);        // the natural table W^k, k < N/2
            for (int r = 0; r < 2; r++) {
                const int bt = q + r * (N >> 2);
                const int grp = bt >> st, k = bt & (half - 1);
                const int a = fft_lds_swizzle((grp << (st + 1)) + k), b = a ^ fft_lds_swizzle(half);
                const cf32 w = twn[k];
                const cf32 uu = x[a];
                const cf32 vv = cmul_tw(x[b], w);
                x[a] = cadd(uu, vv);
                x[b] = csub(uu, vv);
            }
        }
        sync_stage();
    }
    if (wave_local) __syncthreads();                  // the rows are read across waves from here on
}

// CRC of a byte string, MSB-first register, optional reflection (gr::digital::crc semantics)
__device__ uint32_t crc_bits(const uint8_t* data, int len, int nbits, uint32_t poly, uint32_t init,
                             uint32_t final_xor, bool refin, bool refout)
{
    const uint32_t top = 1u << (nbits - 1);
    const uint32_t mask = (nbits == 32) ? 0xFFFFFFFFu : ((1u << nbits) - 1);
    uint32_t reg = init & mask;
    for (int i = 0; i < len; i++) {
        uint32_t byte = data[i];
        if (refin) byte = __brev(byte) >> 24;
        for (int b = 7; b >= 0; b--) {
            const uint32_t inbit = (byte >> b) & 1u;
            const uint32_t msb = (reg & top) ? 1u : 0u;
            reg = (reg << 1) & mask;
            if (msb ^ inbit) reg ^= poly;
        }
    }
    if (refout) reg = __brev(reg) >> (32 - nbits);
    return (reg ^ final_xor) & mask;
}

} // namespace

// ------------------------------------------------------------------------------------------
// K1: Schmidl & Cox timing metric (GR ofdm_sync_sc_cfb as wired at ofdm_receiver.py:71-72).
//   c[j] = r[j] conj(r[j-N/2]),  e[j] = |r[j]|^2      in float (the reference's multiply / mag-squared blocks)
//   P[m] = -sum_{j in (m-N/2, m]} c[j],  E[m] = sum_{j in (m-N, m]} 0.5 e[j]   window sums in DOUBLE
//   metric[m] = |P|^2 / E^2 (float), compared with the threshold
// A 256-thread workgroup owns a tile of 4096 output samples.  The tile plus 1.5 N samples of history is
// staged in LDS once (coalesced 8-byte loads; LDS rows padded by one element per 16 so that lane ==
// strip is bank-conflict free).  Each thread owns a strip of 16 consecutive outputs:
//   pass A  sums c / 0.5e over its strip (values stay in registers) -- the first N/16 lanes also sum
//           one history strip each;
//   scan    inclusive prefix over the strip totals (wave shuffles + 4 wave totals): the window sums
//           just before a strip are differences of this COARSE prefix (3 doubles per 16 samples in
//           LDS instead of 3 per sample);
//   pass C  slides the three window sums along the strip (+ new term from registers, - old term
//           recomputed from the staged samples), forms the metric, records the threshold flags.
// Only the threshold crossings ("edges") leave the chip; plateau logic is sequential (host).
// Without a metric output the test RN(num / den) >= thr is evaluated without the division: with
// B = the rounding boundary between thr and its predecessor (25 significant bits), num >= B * den
// is exact in double (49-bit product) and decides the same way for every finite den > 0.
// ------------------------------------------------------------------------------------------
constexpr int kEqRecipLds = 258;               // frame equaliser: reciprocals of a lane's pilot sample count (<= L + 1)
constexpr int kSyncTile = kSyncTileSamples;   // the stream carry is aligned to the sync tile

// strip length R (outputs per thread): 8 with 512 threads for N <= 512 (half the registers per thread, twice the
// waves in flight), 16 with 256 threads for N = 1024 (the N/R history strips must fit one wave)
template <int R>
__device__ __forceinline__ int sync_pad(int j) { return j + (j >> 4); }

// Inclusive scan of a double over the 64 lanes of a wave with data-parallel-primitive moves (no LDS crossbar, no lane
// selects): four shifts inside the rows of 16 lanes (zero fill at the row start), then the last lane of rows 0 / 2
// added to rows 1 / 3 (row_bcast:15) and lane 31 to the upper half (row_bcast:31).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_scan(double x)
{
    x += dpp_take<0x111, 0xf>(x);   // row_shr:1
    x += dpp_take<0x112, 0xf>(x);   // row_shr:2
    x += dpp_take<0x114, 0xf>(x);   // row_shr:4
    x += dpp_take<0x118, 0xf>(x);   // row_shr:8
    x += dpp_take<0x142, 0xa>(x);   // row_bcast:15 -> rows 1, 3
    x += dpp_take<0x143, 0xc>(x);   // row_bcast:31 -> rows 2, 3
    return x;
}

// PRE: the per-strip interval pre-test (see the comment in front of pass C); CHECK (tools / tests only): every strip
// also takes the exact path, and `check` counts [0] strips, [1] strips the pre-test could not decide, [2] outputs it
// decided differently from the exact test (must stay 0).
template <bool WITH_METRIC, int R, bool PRE = false, bool CHECK = false>
__global__ __launch_bounds__(kSyncTile / R) __attribute__((amdgpu_waves_per_eu(R == 8 ? 6 : 2))) void sync_edges_kernel(const cf32* __restrict__ iq, int64_t n,
                                                                 int64_t m_begin, int N, float thr,
                                                                 unsigned long long* __restrict__ edges,
                                                                 unsigned int* __restrict__ n_edges,
                                                                 unsigned int max_edges,
                                                                 float* __restrict__ metric_out,
                                                                 unsigned long long* __restrict__ check = nullptr)
{
    extern __shared__ double sync_smem[];
    constexpr int kSyncThreads = kSyncTile / R;
    const int half = N / 2;
    const int nh = N / R;                       // history strips in front of the tile
    const int hw = half / R;                    // strips per P window
    const int nS = nh + kSyncThreads;
    const int n_raw = kSyncTile + N + half;     // staged samples: [t0 - N - half, t0 + tile)
    const int64_t t0 = (m_begin / kSyncTile + int64_t(blockIdx.x)) * kSyncTile; // tiles stay aligned to sample 0
    const int64_t b0 = t0 - N - half;
    double* cp_r = sync_smem;                   // coarse inclusive prefix over strips, k = 0 .. nS-1
    double* cp_i = cp_r + nS;
    double* cp_e = cp_i + nS;
    cf32* raw = reinterpret_cast<cf32*>(cp_e + nS);
    __shared__ double wave_tot[3][kSyncThreads / 64];
    __shared__ double halo_tot[3];
    __shared__ unsigned short sflags[kSyncThreads];             // per strip: threshold flags of its R outputs
    [[maybe_unused]] __shared__ unsigned short wl[kSyncThreads]; // PRE: strips the interval pre-test left undecided
    [[maybe_unused]] __shared__ int wl_n;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (PRE && tid == 0) wl_n = 0;                              // (the staging barrier below orders this)

    // Staging: every load of the thread is requested before the first one is waited for (addresses clamped into the
    // stream, values zeroed afterwards -- a bounds test around the load made the compiler run the loop as load, wait,
    // store, load, ...: one exposed memory latency per 512 samples of the tile).
    {
        constexpr int kTileIters = R;                                  // kSyncTile / kSyncThreads
        constexpr int kHistIters = ((R == 8 ? 768 : 1536) + kSyncThreads - 1) / kSyncThreads; // (N + N/2) / threads, largest N served
        cf32 stage[kTileIters + kHistIters];
        // interior tiles (all but the first and the last of a stream) load without any bounds handling; an iteration
        // that lies entirely behind the staged range (N smaller than the largest N served) is skipped by every thread
        const bool interior = b0 >= 0 && b0 + int64_t(kTileIters + kHistIters) * kSyncThreads <= n;
        if (interior) {
#pragma unroll
            for (int it = 0; it < kTileIters + kHistIters; it++)
                if (it * kSyncThreads < n_raw) stage[it] = iq[b0 + tid + it * kSyncThreads];
        } else {
#pragma unroll
            for (int it = 0; it < kTileIters + kHistIters; it++) {
                const int q = tid + it * kSyncThreads;
                const int64_t j = b0 + q;
                const bool inside = q < n_raw && j >= 0 && j < n;
                const cf32 v = iq[inside ? j : 0];
                stage[it] = cf32{ inside ? v.re : 0.0f, inside ? v.im : 0.0f };
            }
        }
#pragma unroll
        for (int it = 0; it < kTileIters + kHistIters; it++) {
            const int q = tid + it * kSyncThreads;
            if (q < n_raw) raw[sync_pad<R>(q)] = stage[it];
        }
    }
    __syncthreads();

    // ---- pass A: strip totals.  Sample j of the strip domain (which starts at t0 - N) is raw[half + j].
    // (the energy terms are formed again in pass C from the staged sample -- one more LDS read per output -- rather
    // than held: R more live registers here meant spills to scratch, i.e. memory round trips, at 80 VGPRs)
    // (the terms are not kept: the exact per-output path forms them again from the two staged samples it reads anyway)
    double tr = 0, ti = 0, te = 0;
    {
        // strips start at multiples of R = 8 or 16 and the padding adds one element per 16: inside a strip the padded
        // index advances with the sample index, so one padded base per row and immediate offsets do
        const int base = half + (nh + tid) * R;
        const cf32* ra = raw + sync_pad<R>(base);
        const cf32* rb = raw + sync_pad<R>(base - half);
#pragma unroll
        for (int s = 0; s < R; s++) {
            if constexpr (!ablated(12)) {
                const cf32 a = ra[s], b = rb[s];
                const float c_re = a.re * b.re + a.im * b.im;
                const float c_im = a.im * b.re - a.re * b.im;
                const float e = 0.5f * (a.re * a.re + a.im * a.im);
                tr += double(c_re); ti += double(c_im); te += double(e);
            }
        }
    }
    double hr = 0, hi = 0, he = 0; // history strip `tid` (wave 0 only: nh <= 64)
    if (tid < nh) {
        const int base = half + tid * R;
        const cf32* ra = raw + sync_pad<R>(base);
        const cf32* rb = raw + sync_pad<R>(base - half);
#pragma unroll
        for (int s = 0; s < R; s++) {
            const cf32 a = ra[s], b = rb[s];
            hr += double(a.re * b.re + a.im * b.im);
            hi += double(a.im * b.re - a.re * b.im);
            he += double(0.5f * (a.re * a.re + a.im * a.im));
        }
    }
    // ---- scans (inclusive, fixed order)
    if (wid == 0) {
        hr = wave_scan(hr); hi = wave_scan(hi); he = wave_scan(he);
        if (lane < nh) { cp_r[lane] = hr; cp_i[lane] = hi; cp_e[lane] = he; }
        if (lane == nh - 1) { halo_tot[0] = hr; halo_tot[1] = hi; halo_tot[2] = he; }
    }
    const double xr = wave_scan(tr), xi = wave_scan(ti), xe = wave_scan(te);
    if (lane == 63) { wave_tot[0][wid] = xr; wave_tot[1][wid] = xi; wave_tot[2][wid] = xe; }
    __syncthreads();
    {
        double br = halo_tot[0], bi = halo_tot[1], be = halo_tot[2];
        for (int w = 0; w < wid; w++) { br += wave_tot[0][w]; bi += wave_tot[1][w]; be += wave_tot[2][w]; }
        cp_r[nh + tid] = br + xr; cp_i[nh + tid] = bi + xi; cp_e[nh + tid] = be + xe;
    }
    __syncthreads();

    // ---- pass C: per strip the window sums just before it (differences of the coarse prefix), then slide
    // threshold test: exact division-free form (see the header comment) or the plain float quotient
    const unsigned thr_bits = unsigned(__float_as_int(thr));
    const bool exact_ok = thr > 1.17549435e-38f && thr < 3.0e38f;
    const double bound = exact_ok ? 0.5 * (double(__int_as_float(int(thr_bits - 1))) + double(thr)) : 0.0;
    const bool tie_up = (thr_bits & 1u) == 0;   // a quotient exactly on the boundary rounds to the even mantissa
    // A window of zeros: the reference's FIR sums are exactly 0 there and 0/0 = NaN is never above the threshold.  The
    // f64 sums of a tile can keep a rounding residue of what the tile held before (<= 2^-40 of the tile's energy after
    // 2 * (tile + history) additions): window energy at or below that is indistinguishable from zero and treated as zero.
    const float e_tile = float(cp_e[nS - 1]);
    const float e_floor = e_tile < 3.0e38f ? e_tile * 0x1p-38f : 0.0f;  // NaN / inf samples in the tile: no floor
    auto above = [&](float num, float den, float E) -> bool {
        // num, den >= 0 (sums of squares) or NaN: one range test covers "finite and den > 0"
        if (!WITH_METRIC && exact_ok && E > e_floor && den > 0.0f && fmaxf(num, den) < 3.0e38f && num == num) {
            const double lhs = double(num), rhs = bound * double(den);
            return tie_up ? lhs >= rhs : lhs > rhs;
        }
        return E > e_floor && num / den >= thr;
    };
    auto metric_parts = [&](double Wr, double Wi, double We, float& num, float& den, float& E) {
        const float Pr = float(-Wr), Pi = float(-Wi);
        E = float(We);
        num = Pr * Pr + Pi * Pi;
        den = E * E;
    };
    // window sums in front of strip `sidx`: nh == 2 hw strips make an E window, hw a P window
    auto window_before = [&](int sidx, double& Wr, double& Wi, double& We) {
        const int km1 = nh + sidx - 1;          // last strip in front of the strip's outputs (>= nh - 1 >= 3)
        Wr = cp_r[km1] - cp_r[km1 - hw];
        Wi = cp_i[km1] - cp_i[km1 - hw];
        We = cp_e[km1] - (sidx > 0 ? cp_e[sidx - 1] : 0.0);
    };
    const int64_t m0 = t0 + int64_t(tid) * R;
    bool prev0 = false;                         // flag of sample m0 - 1 as this tile sees it (lane 0 of the tile uses it)
    if (tid == 0 && m0 - 1 >= 0) {
        double Wr, Wi, We;
        window_before(0, Wr, Wi, We);
        float num, den, E;
        metric_parts(Wr, Wi, We, num, den, E);
        prev0 = above(num, den, E);
    }
    // The exact test of strip `sidx` (any thread can run it: everything comes from LDS): the three window sums slid along
    // the strip in double (+ entering term, - leaving term; the entering term c[m] = r[m] conj(r[m - N/2]) is formed from
    // the two samples the leaving terms need anyway), rounded to float per output.
    auto exact_strip = [&](int sidx) -> unsigned {
        double Wr, Wi, We;
        window_before(sidx, Wr, Wi, We);
        const int rb_new = N + half + sidx * R; // raw index of the strip's first sample
        const cf32* r_new = raw + sync_pad<R>(rb_new);
        const cf32* r_half = raw + sync_pad<R>(rb_new - half);
        const cf32* r_full = raw + sync_pad<R>(rb_new - N);
        unsigned fl = 0;
#pragma unroll
        for (int s = 0; s < R; s++) {
            const cf32 oa = r_half[s], ob = r_full[s];
            const cf32 na = r_new[s];
            const float ncr = na.re * oa.re + na.im * oa.im, nci = na.im * oa.re - na.re * oa.im;
            const float ocr = oa.re * ob.re + oa.im * ob.im, oci = oa.im * ob.re - oa.re * ob.im;
            const float oeh = 0.5f * (ob.re * ob.re + ob.im * ob.im);
            const float neh = 0.5f * (na.re * na.re + na.im * na.im);
            Wr = (Wr + double(ncr)) - double(ocr);
            Wi = (Wi + double(nci)) - double(oci);
            We = (We + double(neh)) - double(oeh);
            float num, den, E;
            metric_parts(Wr, Wi, We, num, den, E);
            if (WITH_METRIC) {
                const int64_t m = t0 + int64_t(sidx) * R + s;
                if (m >= m_begin && m < n) metric_out[m - m_begin] = E > e_floor ? num / den : __int_as_float(0x7fc00000);
            }
            if (above(num, den, E)) fl |= 1u << s;
        }
        return fl;
    };
    // PRE -- an interval test per STRIP in front of the per-output work.  While the windows slide over the strip's R
    // outputs, P moves by at most V = sum of |c| over the terms that enter and leave, and |r_a conj(r_b)| <= e_a + e_b
    // (e = |r|^2 / 2) bounds that by strip sums of the energy terms, which the coarse prefix already holds:
    //   V <= S(k) + 2 S(k - hw) + S(k - 2 hw),   E(s) in [E0 - S(k - 2 hw), E0 + S(k)]     (S(j) = energy sum of strip j)
    // so every output's metric lies in [ max(0, |P0| - V)^2 / Emax^2, (|P0| + V)^2 / Emin^2 ].  If that interval, widened by
    // 1e-4 relative and a slack for the coarse prefix's own rounding (1e-9 of the tile's energy: 10^6 times its
    // error), lies on one side of the threshold, all R flags are known: nearly every strip outside a preamble's plateau
    // and its flanks (the metric sits near 1 / (N / 2) there, or near 1).  The other strips -- and any strip with a
    // window near the zero-window floor, energies whose squares leave the normal range, NaN / inf (comparisons false) --
    // go onto a work list in LDS and are then run through the exact path by the workgroup's first threads, one strip per
    // lane: the few undecided strips of a tile (5 % in a frame stream, spread over half of its waves) cost one pass of
    // one wave instead of one pass of every wave that holds one.  dtl_sync_pretest_check /
    // tests/test_sync_pretest_gpu.py run both tests on every strip of ordinary and adversarial streams: 0 disagreements.
    // A tile that holds a term that is not finite (a NaN / inf sample, or samples whose products overflow): the prefix
    // sums are useless from that term on, but the reference's FIR windows are not -- only the outputs whose OWN window
    // holds such a term are lost (their metric is NaN, never above the threshold).  Such a tile (tile-uniform test on the
    // three prefix totals; rare) takes this path instead: per strip the window in front of it is summed afresh over its
    // finite terms, with counts of the terms that are not finite and of the energy terms that are not zero, and slid
    // along the strip with the same bookkeeping (the oracle's orc_sync_sc keeps the same counts on its running sums).
    const bool odd_tile = !(fabs(cp_e[nS - 1]) < 3.0e38) || !(fabs(cp_r[nS - 1]) < 3.0e38) || !(fabs(cp_i[nS - 1]) < 3.0e38);
    auto odd_strip = [&](int sidx) -> unsigned {                       // bits 0 .. R-1: the flags; bit 31: the output in front
        const int rb_new = N + half + sidx * R;                       // raw index of the strip's first sample
        auto at = [&](int q) -> cf32 { return raw[sync_pad<R>(q)]; };
        double Wr = 0, Wi = 0, We = 0;
        int bad_c = 0, bad_e = 0, nz_e = 0;
        auto fin = [](float x) -> bool { return fabsf(x) <= 3.4028234e38f; };
        auto c_term = [&](int q, int sign) {                          // c of the sample at raw index q
            const cf32 a = at(q), b = at(q - half);
            const float cr = a.re * b.re + a.im * b.im, ci = a.im * b.re - a.re * b.im;
            if (fin(cr) && fin(ci)) { Wr += sign * double(cr); Wi += sign * double(ci); }
            else bad_c += sign;
        };
        auto e_term = [&](int q, int sign) {
            const cf32 a = at(q);
            const float e = 0.5f * (a.re * a.re + a.im * a.im);
            if (fin(e)) { We += sign * double(e); nz_e += e != 0.0f ? sign : 0; }
            else bad_e += sign;
        };
        for (int k = 1; k <= half; k++) c_term(rb_new - k, +1);
        for (int k = 1; k <= N; k++) e_term(rb_new - k, +1);
        float metric = 0.0f;
        auto flag_now = [&](void*) -> bool {
            float num, den, E;
            metric_parts(Wr, Wi, nz_e > 0 ? We : 0.0, num, den, E);
            const bool clean = bad_c == 0 && bad_e == 0;
            metric = clean && E > 0.0f ? num / den : __int_as_float(0x7fc00000);
            return clean && E > 0.0f && num / den >= thr;
        };
        unsigned fl = flag_now(nullptr) ? 0x80000000u : 0u;
        for (int s = 0; s < R; s++) {
            c_term(rb_new + s, +1); c_term(rb_new + s - half, -1);
            e_term(rb_new + s, +1); e_term(rb_new + s - N, -1);
            if (flag_now(nullptr)) fl |= 1u << s;
            if (WITH_METRIC) {
                const int64_t m = t0 + int64_t(sidx) * R + s;
                if (m >= m_begin && m < n) metric_out[m - m_begin] = metric;
            }
        }
        return fl;
    };
    unsigned flags = 0;
    if constexpr (ablated(11) || ablated(12)) {
        sflags[tid] = 0;
        __syncthreads();
    } else if (odd_tile) {
        const unsigned fl = odd_strip(tid);
        sflags[tid] = (unsigned short)fl;
        if (tid == 0) prev0 = m0 - 1 >= 0 && (fl >> 31) != 0;
        __syncthreads();
    } else if constexpr (PRE && !WITH_METRIC) {
        const int k = nh + tid;                     // this strip
        double Wr, Wi, We;
        window_before(tid, Wr, Wi, We);
        const float s_new = float(cp_e[k] - cp_e[k - 1]);
        const float s_half = float(cp_e[k - hw] - cp_e[k - hw - 1]);
        const float s_old = float(cp_e[tid] - (tid > 0 ? cp_e[tid - 1] : 0.0));
        const float p0r = float(Wr), p0i = float(Wi), e0 = float(We);
        const float slack = e_tile * 1.0e-9f;
        const float V = (s_new + 2.0f * s_half + s_old) * 1.0001f + slack;
        const float pabs = sqrtf(p0r * p0r + p0i * p0i);
        const float e_max = (e0 + s_new) * 1.0001f + slack, e_min = (e0 - s_old) * 0.9999f - slack;
        const float up = pabs * 1.0001f + V, lo = fmaxf(0.0f, pabs * 0.9999f - V);
        const bool ok = exact_ok && thr < 1.0e6f && e_min > fmaxf(fmaxf(1.0e-6f * e_tile, 2.0f * e_floor), 1.0e-15f) && e_max < 1.0e15f;
        const bool all_below = ok && up * up < thr * 0.9999f * (e_min * e_min);
        const bool all_above = ok && lo * lo > thr * 1.0001f * (e_max * e_max);
        const bool decided = all_below || all_above;
        const unsigned pre_flags = all_above ? ((1u << R) - 1u) : 0u;
        if constexpr (CHECK) {
            const unsigned ex = exact_strip(tid);
            if (m0 < n) {
                atomicAdd(&check[0], 1ull);
                if (!decided) atomicAdd(&check[1], 1ull);
                else if (ex != pre_flags) atomicAdd(&check[2], (unsigned long long)__popc(ex ^ pre_flags));
            }
            sflags[tid] = (unsigned short)ex;
            __syncthreads();
        } else {
            if (decided) sflags[tid] = (unsigned short)pre_flags;
            else wl[atomicAdd(&wl_n, 1)] = (unsigned short)tid;
            __syncthreads();
            const int n_wl = wl_n;
            for (int w = tid; w < n_wl; w += kSyncThreads) {
                const int sidx = wl[w];
                sflags[sidx] = (unsigned short)exact_strip(sidx);
            }
            __syncthreads();
        }
    } else {
        sflags[tid] = (unsigned short)exact_strip(tid);
        __syncthreads();
    }
    flags = sflags[tid];
    // (a thread that goes on -- m0 < n -- has a complete strip in front of it: its last output is bit R - 1)
    const bool prev = tid ? ((sflags[tid - 1] >> (R - 1)) & 1u) != 0 : prev0;
    if (m0 >= n) return;
    // crossings of the strip: bit s of `trans` = flag(s) != flag(s - 1); nearly every strip has none
    const int n_valid = int(min<int64_t>(R, n - m0));
    const int first_valid = int(max<int64_t>(0, m_begin - m0));
    unsigned valid = n_valid >= 32 ? 0xFFFFFFFFu : ((1u << n_valid) - 1u);
    valid &= first_valid >= 32 ? 0u : ~((1u << first_valid) - 1u);
    unsigned trans = (flags ^ ((flags << 1) | (prev ? 1u : 0u))) & valid;
    while (trans) {
        const int sidx = __ffs(int(trans)) - 1;
        trans &= trans - 1;
        const bool cur = (flags >> sidx) & 1u;
        const unsigned int slot = atomicAdd(n_edges, 1u);
        if (slot < max_edges) edges[slot] = (static_cast<unsigned long long>(m0 + sidx) << 1) | (cur ? 1ull : 0ull);
    }
}

// K1b: arg(P) at each raw trigger (complex_to_arg + sample_and_hold_ff), one wave per trigger.
__global__ __launch_bounds__(64) void trigger_angle_kernel(const cf32* __restrict__ iq, int64_t n, int N,
                                                          const int64_t* __restrict__ trig, int n_trig,
                                                          float* __restrict__ angle)
{
    const int half = N / 2;
    const int w = blockIdx.x;
    if (w >= n_trig) return;
    const int64_t pos = trig[w];
    double pr = 0, pi = 0;
    for (int m = threadIdx.x; m < half; m += 64) {
        const int64_t a_i = pos - m, b_i = pos - m - half;
        cf32 a = { 0, 0 }, b = { 0, 0 };
        if (a_i >= 0 && a_i < n) a = iq[a_i];
        if (b_i >= 0 && b_i < n) b = iq[b_i];
        pr += double(a.re * b.re + a.im * b.im);
        pi += double(a.im * b.re - a.re * b.im);
    }
    for (int off = 32; off > 0; off >>= 1) {
        pr += __shfl_down(pr, off);
        pi += __shfl_down(pi, off);
    }
    // atan2 in double, rounded to float once (same definition as the CPU chain; libm-independent)
    if (threadIdx.x == 0) angle[w] = float(atan2(double(float(-pi)), double(float(-pr))));
}

// ------------------------------------------------------------------------------------------
// Phase A (K2..K5): per trigger -- cut (2+h) symbols, FFT, ofdm_chanest_vcvc, header equaliser
// (ofdm_frame_equalizer_vcvc around ofdm_adaptive_header_equalizer, alpha == 1), serializer,
// BPSK slicer, ofdm_adaptive_packet_header::header_parser.
// ------------------------------------------------------------------------------------------
// STAGE == true serves the per-block entry points dtl_chanest / dtl_header_decode: the rows come from frequency-domain
// vectors (what fft_vcc hands the reference's ofdm_chanest_vcvc / header equaliser) instead of being cut out of the
// IQ stream and transformed here; everything behind the rows is the same code.
// Registers: held to 64 per lane (8 waves per SIMD) -- a 1024-point workgroup is 16 waves, and with the 72 registers the
// compiler takes unasked only ONE such workgroup fits a CU (20 waves); with two resident, one's barriers and sample
// loads hide under the other's arithmetic (c5 header kernel 6.6 -> 5.0 ms per step; no scratch either way).
template <bool STAGE>
__global__ __attribute__((amdgpu_waves_per_eu(8, 8))) void header_kernel(RxTables t, const cf32* __restrict__ iq, int64_t n_samples,
                              const Candidate* __restrict__ cands, const CfoSegment* __restrict__ segs,
                              int n_segs, HeaderRec* __restrict__ recs, cf32* __restrict__ taps_out,
                              uint8_t* __restrict__ hdr_bits_out, int search_chunk, HeaderStageIo io)
{
    extern __shared__ unsigned char smem_raw[];
    const int N = t.fft_len, h = t.header_len, tid = threadIdx.x;
    cf32* Y = reinterpret_cast<cf32*>(smem_raw);            // (2+h) x N, natural (unshifted) order
    cf32* H = Y + size_t(2 + h) * N;                        // N taps (shifted positions)
    float* gabs = reinterpret_cast<float*>(H + N);          // N/2+2 |B(g)| values
    uint8_t* bits = reinterpret_cast<uint8_t*>(gabs + N / 2 + 2); // hdr_bits
    cf32* tw = reinterpret_cast<cf32*>(bits + ((t.hdr_bits + 15) & ~15)); // N per-pass twiddles
    cf32* terms = tw + N;                                // n_g x (search_chunk + 1) products of the offset search
    cf32* s_cvc = terms + size_t((t.g_max - t.g_min) / 2 + 1) * (search_chunk + 1);   // n_sync_act: conj(corr_v) of the participating carriers
    uint16_t* s_kact = reinterpret_cast<uint16_t*>(s_cvc + t.n_sync_act);              // n_sync_act: their positions
    __shared__ int s_off;

    tw[tid] = t.twiddle[tid];
    if constexpr (STAGE) {
        // rows io.row0 .. io.row0 + io.n_rows - 1 from the caller's shifted vectors (DC at N/2); the others are zero
        const int nat = fft_lds_swizzle((tid + N / 2) & (N - 1));       // where shifted position tid sits in a row
        for (int r = 0; r < 2 + h; r++) {
            cf32 v = { 0.0f, 0.0f };
            if (r >= io.row0 && r < io.row0 + io.n_rows) v = io.freq_in[(size_t(blockIdx.x) * io.n_rows + (r - io.row0)) * N + tid];
            Y[size_t(r) * N + nat] = v;
        }
        __syncthreads();
    } else {
        const Candidate cd = cands[blockIdx.x];
        const SegWindow sw = seg_window(segs, n_segs, cd.seg);
        for (int s = 0; s < 2 + h; s += 4)
            load_and_fft4<(kAblate == 42 || kAblate == 43) ? kAblate : 0>(t, iq, n_samples, cd.trigger, s, min(4, 2 + h - s), segs, n_segs, sw, Y + size_t(s) * N, tw);
    }
    const bool given_taps = STAGE && io.taps_in != nullptr;     // dtl_header_decode: channel estimate supplied
    // shifted view: Ysh(s, k) = Y[s][(k + N/2) % N] (rows are stored swizzled)
    auto ysh = [&](int s, int k) -> cf32 { return Y[size_t(s) * N + fft_lds_swizzle((k + N / 2) & (N - 1))]; };

    // get_carr_offset: B(g) = sum_k conj(y1[k+g]) * conj(corr_v[k]) * y2[k+g], first maximum wins.
    // The reference adds the terms in carrier order in float, so the sum of one offset g is a serial chain; what can be
    // shared is the work inside it.  Lane g < n_g owns the running sum of offset g (a register, carried over the rounds);
    // the products are formed by ALL lanes, a chunk of `search_chunk` participating carriers at a time (every offset's
    // products of the chunk parked in LDS: n_g x (chunk + 1) entries, the odd row stride keeps the adding lanes' 64-bit
    // reads on distinct banks), then the n_g lanes add their chunk in order.  The chunk is chosen on the host so that the
    // product table stays small whatever the transform length: a 1024-point header searches 97 offsets over 416 carriers
    // (323 KB of products -- the form that parked them all fell back to one lane per offset forming its own 1024 products:
    // 50 us per header, 23 ms per 28 k-frame step).
    const int n_g = (t.g_max - t.g_min) / 2 + 1;
    const int n_act = t.n_sync_act;
    if (given_taps) {
        // (a caller's offset: at or beyond +-h N every shifted position falls outside the header, and so does it for any
        // larger value -- held to that range so that no later sum or negation of it can leave the int range)
        if (tid == 0) s_off = min(max(io.carr_off_in[blockIdx.x], -h * N), h * N);
    } else if (ablated(41)) {
        if (tid < n_g) gabs[tid] = 0.0f;
    } else {
        const int CH = search_chunk, ch_log = __ffs(CH) - 1;          // a power of two
        for (int a = tid; a < n_act; a += blockDim.x) {               // the participating carriers and conj(corr_v) of each
            const int k = t.sync_act[a];
            s_kact[a] = uint16_t(k);
            s_cvc[a] = cconj(t.corr_v[k]);
        }
        __syncthreads();
        cf32 acc = { 0.0f, 0.0f };
        for (int a0 = 0; a0 < n_act; a0 += CH) {
            const int na = min(CH, n_act - a0);
            for (int p = tid; p < (n_g << ch_log); p += blockDim.x) {
                const int gi = p >> ch_log, a = p & (CH - 1);
                if (a < na) {
                    const int k = s_kact[a0 + a], g = t.g_min + 2 * gi;
                    terms[gi * (CH + 1) + a] = cmul(cmul(cconj(ysh(0, k + g)), s_cvc[a0 + a]), ysh(1, k + g));
                }
            }
            __syncthreads();
            if (tid < n_g) {
                // eight terms at a time: the loads go out together (one LDS latency per eight, not per term), the adds follow
                // in carrier order; a slot behind the chunk's last carrier is read (in bounds: the row has CH + 1 entries)
                // and not added
                const cf32* row = terms + tid * (CH + 1);
                for (int i0 = 0; i0 < na; i0 += 8) {
                    cf32 v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = row[min(i0 + j, CH)];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const cf32 sum = cadd(acc, v[j]);
                        acc = i0 + j < na ? sum : acc;
                    }
                }
            }
            __syncthreads();                                 // the rows are rewritten by the next round
        }
        if (tid < n_g) gabs[tid] = sqrtf(acc.re * acc.re + acc.im * acc.im);
    }
    __syncthreads();
    if (tid == 0 && !given_taps) {
        int best = 0;
        float bmax = 0.0f;
        for (int gi = 0; gi < n_g; gi++)
            if (gabs[gi] > bmax) { bmax = gabs[gi]; best = t.g_min + 2 * gi; }
        s_off = best;
    }
    __syncthreads();
    const int off = s_off;
    // get_chan_taps
    if (given_taps) {
        H[tid] = io.taps_in[size_t(blockIdx.x) * N + tid];
    } else {
        const int j = tid; // tap index
        cf32 tap = { 0.0f, 0.0f };
        const int lo = off > 0 ? 0 : -off, hi = off > 0 ? N - off : N;
        const cf32 r = t.ref_sym[j];
        if (j >= lo && j < hi && cnonzero(r)) tap = cdiv(ysh(1, j + off), r);
        H[j] = tap;
    }
    if constexpr (STAGE) {
        // ofdm_chanest_vcvc's own outputs: the taps of the "ofdm_sync_chan_taps" tag and the carrier offset
        if (io.taps_raw_out) io.taps_raw_out[size_t(blockIdx.x) * N + tid] = H[tid];
        if (io.carr_off_out && tid == 0) io.carr_off_out[blockIdx.x] = off;
        if (!io.do_header) return;
    }
    __syncthreads();
    // header symbols: integer-CFO shift on the flattened frame, phase derotation, one-tap equalise
    const int total = h * N;
    int bit_base = 0;
    for (int i = 0; i < h; i++) {
        const int set = i % t.n_occ_sets;
        const int nocc = t.occ_len[set];
        for (int r = tid; r < nocc; r += blockDim.x) {
            const int k = t.occ_idx[set * t.max_occ + r];
            const int p = i * N + k, src = p + off;
            cf32 v = { 0.0f, 0.0f };
            if (off < 0 ? (p >= -off) : (p < total - off)) v = ysh(2 + src / N, src % N);
            const float ph = float(-(2.0 * kPi) * off * t.cp_len / N * (i + 1));
            v = cmul(v, expj(ph));
            const cf32 eq = cdiv(v, H[k]);
            bits[bit_base + r] = eq.re > 0 ? 1 : 0; // constellation_bpsk::decision_maker
        }
        bit_base += nocc;
    }
    __syncthreads();
    // taps handed to the payload equaliser: state after the header, rotated (GR :"Update the
    // channel state regarding the frequency offset")
    {
        const float ph = float((2.0 * kPi) * off * t.cp_len / N * h);
        taps_out[size_t(blockIdx.x) * N + tid] = cmul(H[tid], expj(ph));
    }
    if (hdr_bits_out)
        for (int b = tid; b < t.hdr_bits; b += blockDim.x) hdr_bits_out[size_t(blockIdx.x) * t.hdr_bits + b] = bits[b];
    // header_parser (packet_header.cc:231-312), bits_per_byte == 1.  The work is spread over the lanes of wave 0
    // (one lane used to walk ~900 LDS bytes and 400 CRC bit steps while the workgroup's LDS stayed allocated):
    // the header bits become ballot words, the CRC input bytes are packed one per lane, the CRC-16 register runs
    // byte-wise through a 256-entry table built by all threads.
    const int HL = t.hdr_bits;
    uint16_t* crc_tab = reinterpret_cast<uint16_t*>(Y);                 // the FFT rows are dead: 256 table entries,
    uint8_t* crc_buf = reinterpret_cast<uint8_t*>(Y) + 512;             // then the CRC input bytes
    int len = (HL - 16) / 8;
    if ((HL - 16) % 8) ++len;
    const bool table_fits = 512 + size_t(len > 0 ? len : 0) <= size_t(2 + h) * N * sizeof(cf32);
    if (table_fits) {
        for (int e = tid; e < 256; e += blockDim.x) { // MSB-first CRC-16, polynomial 0x1021: register after byte e
            uint32_t reg = uint32_t(e) << 8;
            for (int b = 0; b < 8; b++) reg = (reg & 0x8000u) ? ((reg << 1) ^ 0x1021u) : (reg << 1);
            crc_tab[e] = uint16_t(reg);
        }
        // pack_crc (:93-105) keeps the reference's loop bound i*len+j < header_len
        for (int i = tid; i < len; i += blockDim.x) {
            uint8_t v = 0;
            for (int j = 0; j < 8 && i * len + j < HL; j++) v = uint8_t((v << 1) | (bits[i * 8 + j] & 1));
            crc_buf[i] = v;
        }
    }
    __syncthreads();
    if (tid < 64 && !ablated(44)) {
        // header bit k (k < 128 covers every field and the CRC of both header layouts) as bit k of two ballots
        const int f_bits = t.has_fec ? 80 : 32;
        const unsigned long long w0 = __ballot(tid < HL && (bits[tid < HL ? tid : 0] & 1));
        const unsigned long long w1 = __ballot(tid + 64 < HL && (bits[tid + 64 < HL ? tid + 64 : 0] & 1));
        // the 16 CRC bits follow the fields (positions f_bits .. f_bits+15, cut at HL)
        if (tid == 0) {
            HeaderRec r{};
            int k = 0;
            auto field = [&](int nb) {
                int v = 0;
                for (int i = 0; i < nb && k < HL; i++, k++) v |= int(((k < 64 ? w0 >> k : w1 >> (k - 64)) & 1ull)) << i;
                return v;
            };
            r.payload_len = field(12);
            r.frame_no = field(12);
            r.cnst = field(4);
            r.fb_cnst = field(4);
            if (t.has_fec) {
                r.tb_no = field(12); r.fec_fb = field(4); r.tb_offset = field(12); r.fec = field(4); r.tb_payload = field(16);
            }
            (void)f_bits;
            uint32_t crc;
            if (table_fits) {
                uint32_t reg = 0xFFFFu;
                for (int i = 0; i < len; i++) reg = ((reg << 8) & 0xFFFFu) ^ crc_tab[((reg >> 8) ^ crc_buf[i]) & 0xFFu];
                crc = __brev(reg) >> 16;                               // refout, no final xor
            } else {
                for (int i = 0; i < len; i++) {
                    uint8_t v = 0;
                    for (int j = 0; j < 8 && i * len + j < HL; j++) v = uint8_t((v << 1) | (bits[i * 8 + j] & 1));
                    crc_buf[i] = v;
                }
                crc = crc_bits(crc_buf, len, 16, 0x1021u, 0xFFFFu, 0u, false, true);
            }
            int ok = 1;
            for (int i = 0; i < 16 && k < HL; i++, k++)
                if (uint32_t((k < 64 ? w0 >> k : w1 >> (k - 64)) & 1ull) != ((crc >> i) & 1u)) ok = 0;
            r.crc_ok = ok;
            r.carr_offset = off;
            recs[blockIdx.x] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Phase B (K2,K3,K6,K7,K8): per accepted frame -- cut L payload symbols, FFT,
// ofdm_adaptive_frame_equalizer_vcvc_impl::work (:152-193) + frame_equalize
// (ofdm_adaptive_equalizer.cc:217-268) + mpsk_snr_est_simple, ofdm_serializer_vcc and
// ofdm_adaptive_constellation_soft_cf (soft_cf_impl.cc:89-148) or the hard demapper of the
// no-FEC tail.
// ------------------------------------------------------------------------------------------
// constellation::calc_soft_dec (exact MAP over all M points, accumulation in point order) followed by
// the LSB-first reversal of soft_cf_impl.cc:146.  BPS is a compile-time constant so that the bit tests
// of the accumulation vanish; the arithmetic per point (dist, -dist/npwr, exp, adds in point order) is
// the reference's.  The single-precision quotient -dist/npwr has one divisor per frame: its correctly rounded
// reciprocal is formed once (float(1 / double(npwr)): a quotient of two floats is never closer than 2^-49
// relative to a rounding boundary, the f64 quotient is within 2^-53) and every quotient takes the refinement
// half of the IEEE division sequence (dm_neg_div_by: 1 multiply + 4 FMAs instead of ~11 instructions);
// quotients in the subnormal range all map to exp() == 1.  tools/math_check.hip checks both claims.
// MODE 0: squared distance / npwr (default), 1: plain distance / npwr, 2: FAST arithmetic (DTL_DEMAP_FAST; squared
// distances): 16-QAM in the separable form -- exp(-(dx^2 + dy^2)/npwr) factors, the other axis' factor cancels in
// P(bit = 1) / P(bit = 0), so a bit needs the 4 levels of its own axis: 8 exp + 8 log instead of 16 + 8 (oracle:
// calc_soft_dec_qam16_fast; NOT the reference's bits).  Other constellations: as mode 0.
// o[j] = log(p1[j]) - log(p0[j]); ORD: without the logarithm's special cases when every sum of the wave is a normal number
// (the sums are sums of exponentials <= 1: finite; a bit value whose nearest point is ~87 noise scales away underflows)
template <int BPS, bool ORD>
__device__ __forceinline__ void demap_log_ratios(const float (&p1)[BPS], const float (&p0)[BPS], float* __restrict__ o)
{
    if constexpr (ORD) {
        float least = p1[0];
#pragma unroll
        for (int j = 0; j < BPS; j++) least = fminf(least, fminf(p1[j], p0[j]));
        if (__all(least >= 1.17549435e-38f)) {
#pragma unroll
            for (int j = 0; j < BPS; j++) o[j] = ablated(32) ? p1[j] - p0[j] : dm_log<true>(p1[j]) - dm_log<true>(p0[j]);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < BPS; j++) o[j] = dm_log(p1[j]) - dm_log(p0[j]);
}

// the per-lane half of soft_demap_symbol's ORD precondition (the caller takes the ORD form when it holds for every
// lane of the wave): |re|, |im| <= 1e4 and 1e-20 <= npwr <= 1e20 -- a distance (squared or not) is then below 3e8, its
// quotient by npwr below 3e28, and neither is NaN
__device__ __forceinline__ bool demap_operands_ordinary(cf32 s, float npwr)
{
    return fabsf(s.re) <= 1.0e4f && fabsf(s.im) <= 1.0e4f && npwr >= 1.0e-20f && npwr <= 1.0e20f;
}

// ORD ("ordinary operands", decided per wave by the caller: demap_operands_ordinary): the symbol is finite and of
// moderate size and the noise scale is an ordinary positive number, so every exponent is a finite number <= 0 and the
// quotient guard, the exponential's overflow / NaN selects -- and, when all the sums of a wave turn out to be normal
// numbers (checked here), the logarithm's rescaling and special results -- would select nothing: the same operations
// in the same order without them, a quarter fewer instructions.
template <int BPS, int MODE, bool ORD = false>
__device__ __forceinline__ void soft_demap_symbol(cf32 s, const cf32* __restrict__ pts, float npwr, float rinv,
                                                  float* __restrict__ o)
{
    constexpr int M = 1 << BPS;
    constexpr bool PLAIN_DIST = MODE == 1;
    if constexpr (BPS == 4 && MODE == 2) {
        float er[4], ei[4];
        if constexpr (ORD) {
            dm_f2 d2[4], ex[4];                           // (real level, imaginary level) pairs through the packed instructions
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float dr = s.re - pts[q << 1].re, di = s.im - pts[(q & 1) | ((q >> 1) << 3)].im;
                d2[q] = dm_f2{ dr * dr, di * di };
            }
            dm_exp_neg_div_by_pairs<4>(d2, npwr, rinv, ex);
#pragma unroll
            for (int q = 0; q < 4; q++) { er[q] = ex[q].x; ei[q] = ex[q].y; }
        } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float dr = s.re - pts[q << 1].re;                        // points 0, 2, 4, 6: the four real levels
            const float di = s.im - pts[(q & 1) | ((q >> 1) << 3)].im;     // points 0, 1, 8, 9: the four imaginary levels
            er[q] = dm_exp<ORD>(dm_neg_div_by<ORD>(dr * dr, npwr, rinv));
            ei[q] = dm_exp<ORD>(dm_neg_div_by<ORD>(di * di, npwr, rinv));
        }
        }
        // index bit 1 / 2 = LSB / MSB of the real level, bit 0 / 3 = LSB / MSB of the imaginary level; LSB first
        const float a1[4] = { ei[1] + ei[3], er[1] + er[3], er[2] + er[3], ei[2] + ei[3] };
        const float a0[4] = { ei[0] + ei[2], er[0] + er[2], er[0] + er[1], ei[0] + ei[1] };
        demap_log_ratios<4, ORD>(a1, a0, o);
        return;
    }
    float p0[BPS], p1[BPS];
    cf32 lp[M];
#pragma unroll
    for (int m = 0; m < M; m++) lp[m] = pts[m];
#pragma unroll
    for (int j = 0; j < BPS; j++) { p0[j] = 0.0f; p1[j] = 0.0f; }
    // 16-QAM is a 4 x 4 grid (real part chosen by index bits 1-2, imaginary part by bits 0 and 3): the sixteen
    // squared component distances are four + four distinct values; computing each once gives the same floats
    float dr2[4], di2[4];
    if (BPS == 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float dr = s.re - lp[q << 1].re;                        // points 0, 2, 4, 6: the four real levels
            const float di = s.im - lp[(q & 1) | ((q >> 1) << 3)].im;     // points 0, 1, 8, 9: the four imaginary levels
            dr2[q] = dr * dr;
            di2[q] = di * di;
        }
    }
    auto distance = [&](int m) -> float {
        float sq;
        if (BPS == 4) {
            sq = dr2[(m >> 1) & 3] + di2[(m & 1) | ((m >> 3) << 1)];
        } else {
            const float dr = s.re - lp[m].re, di = s.im - lp[m].im;
            sq = dr * dr + di * di;
        }
        // compile-time switch: a runtime select would evaluate the correctly rounded sqrtf for every point
        return PLAIN_DIST ? sqrtf(sq) : sq;
    };
    if constexpr (ORD && M >= 2) {
        // two points per step through the packed instructions (same operations per point, same order of the sums)
        constexpr int K = M >= 8 ? 4 : M / 2;            // pairs per step
#pragma unroll
        for (int m0 = 0; m0 < M; m0 += 2 * K) {
            dm_f2 dist2[K], d[K];
#pragma unroll
            for (int i = 0; i < K; i++) dist2[i] = dm_f2{ distance(m0 + 2 * i), distance(m0 + 2 * i + 1) };
            if constexpr (ablated(31)) {
#pragma unroll
                for (int i = 0; i < K; i++) d[i] = dist2[i] * dm_f2{ rinv, rinv };
            } else {
                dm_exp_neg_div_by_pairs<K>(dist2, npwr, rinv, d);
            }
#pragma unroll
            for (int i = 0; i < K; i++) {
                const int m = m0 + 2 * i;
#pragma unroll
                for (int j = 0; j < BPS; j++) {
                    if ((m >> j) & 1) p1[j] += d[i].x;
                    else p0[j] += d[i].x;
                    if (((m + 1) >> j) & 1) p1[j] += d[i].y;
                    else p0[j] += d[i].y;
                }
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < M; m++) {
            const float d = dm_exp<ORD>(dm_neg_div_by<ORD>(distance(m), npwr, rinv)); // -dist / npwr exactly; fixed-sequence expf (dtl_math.h)
#pragma unroll
            for (int j = 0; j < BPS; j++) {
                if ((m >> j) & 1) p1[j] += d;
                else p0[j] += d;
            }
        }
    }
    demap_log_ratios<BPS, ORD>(p1, p0, o);
}

// dist_mode argument (0 / 1, + DTL_DEMAP_FAST) -> MODE of soft_demap_symbol; the fast form exists for squared distances
static inline int demap_mode(int dist_mode)
{
    const int base = (dist_mode & 0xff) ? 1 : 0;
    return ((dist_mode & 0x100) && base == 0) ? 2 : base;
}

struct Welford {
    double n, mean, m2;
};

// recip[k] = RN(1 / k): the counts are small integers, so their divisions take the table form (dm_div_by_count)
// (the frame equaliser passes an LDS copy of the table's first `n_lds` entries: a global load in every level of its
// merge tree would be a memory latency each)
__device__ __forceinline__ Welford welford_merge(Welford a, Welford b, const double* __restrict__ recip,
                                                 const double* recip_lds = nullptr, int n_lds = 0)
{
    if (b.n == 0) return a;
    if (a.n == 0) return b;
    const double n = a.n + b.n, d = b.mean - a.mean;
    double r;
    if (int(n) < n_lds) r = recip_lds[int(n)];
    else r = recip[int(n)];
    return { n, a.mean + d * dm_div_by_count(b.n, n, r), a.m2 + b.m2 + d * d * dm_div_by_count(a.n * b.n, n, r) };
}

// Phase B is two kernels.  frame_eq_kernel needs barriers and LDS (FFT) but no more than four symbols of it:
// the frame equaliser is sequential over the symbols of ONE carrier, so a lane can equalise the four
// symbols of a batch as soon as they are transformed, and nothing forces the frame to stay resident --
// 9 KB of LDS per workgroup instead of 49 KB, eight workgroups per CU instead of three.  The soft demapper
// needs the noise estimate of the whole frame and therefore runs afterwards, as a plain element-wise kernel
// at full occupancy (demap_kernel); the equalised symbols cross HBM once in between (8 B per data symbol).
//
// Integer carrier offset (`off`, :154-166 of the frame equaliser): destination (symbol i, carrier k) takes
// the FFT output at flattened position i N + k + off, zero outside the frame.  For a fixed carrier k the
// source is always column (k + off) mod N of row i + lag with ONE lag in {-1, 0, +1}: the lane walks the
// source rows in order and equalises destination symbol (row - lag); the zero symbol comes first (lag -1)
// or last (lag +1).
// STAGED (the usual configuration: one data-carrier set, at most 32 pilot-carrier sets, no eq_hard / eq_soft rows
// requested): an equaliser step touches no global memory at all -- the pilot values of a batch's steps are fetched
// into LDS before the batch, the equalised data symbols are parked in LDS and stored after it.  The general form
// requests its table entries one step ahead and stores from inside the step; the compiler then drains the memory
// queue once per step (loads and stores share one in-order counter), the staged form once per batch.
// FREQ_IN == true serves the per-block entry point dtl_equalize_frames: `iq` then holds, per frame, the L x N
// frequency-domain vectors the reference's frame equaliser block receives from fft_vcc (shifted, DC at N/2) and the
// batches are copied from there instead of being cut out of the IQ stream and transformed.
// DEMAP >= 0 (opt-in, see payload_is_fused; DEMAP = MODE of soft_demap_symbol): the soft demapper runs in
// this kernel's epilogue instead of a kernel of its own, and the equalised data symbols never cross HBM.  The demapper
// needs the frame's FINAL noise estimate, so the frame's equalised data symbols stay on chip until the Welford merge
// has finished: the first `rho0 + 1` OFDM symbols parked in LDS in serializer order (as many as leave room for the four
// workgroups per CU the register budget allows), the rest in REGISTERS of the lane that equalised them (2 VGPRs per
// symbol, kEqRegSlots slots: slot = source row - rho0, a wave-uniform index whatever the lane's own lag; rho0 a multiple
// of 4).  Then ALL N lanes pick symbols in serializer order (lane == data symbol, not carrier: no lane idles on a
// pilot or an unused carrier) -- from the parked rows directly, from the registers through the (now free) FFT scratch in
// batches of four OFDM symbols -- and store their soft bits at llr_off[frame].  The price is registers: four workgroups
// per CU instead of five for the sequential equaliser recursion; the gain is that the demapper's arithmetic fills issue
// slots the recursion leaves idle, and 8 B per data symbol less in each direction.
constexpr int kEqRegSlots = 8;
constexpr int kEqFusedMaxSyms = 28;             // frame_length served by the fused form

// keep[4 B + OFF + j] <-> v[j], j = 0 .. J-1, for a wave-uniform B, written as selects over STATIC register indices
// (a switch over B is turned into one indexed access by the optimiser, which sends the array to scratch memory)
template <bool STORE, int OFF, int J>
__device__ __forceinline__ void keep_window(int B, float (&kr)[kEqRegSlots], float (&ki)[kEqRegSlots], cf32 (&v)[J])
{
    if constexpr (STORE) {
#pragma unroll
        for (int s = 0; s < kEqRegSlots; s++) {
#pragma unroll
            for (int j = 0; j < J; j++) {
                // slot s takes v[j] when 4 B + OFF + j == s
                if ((s - OFF - j) % 4 == 0 && s - OFF - j >= 0) {
                    const bool hit = B == (s - OFF - j) / 4;
                    kr[s] = hit ? v[j].re : kr[s];
                    ki[s] = hit ? v[j].im : ki[s];
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < J; j++) {
            float re = 0.0f, im = 0.0f;
#pragma unroll
            for (int s = 0; s < kEqRegSlots; s++) {
                if ((s - OFF - j) % 4 == 0 && s - OFF - j >= 0) {
                    const bool hit = B == (s - OFF - j) / 4;
                    re = hit ? kr[s] : re;
                    im = hit ? ki[s] : im;
                }
            }
            v[j] = cf32{ re, im };
        }
    }
}

template <bool STAGED, bool FREQ_IN = false, int DEMAP = -1>
__global__ __attribute__((amdgpu_waves_per_eu(DEMAP >= 0 ? 4 : 5, DEMAP >= 0 ? 4 : 5))) void frame_eq_kernel(RxTables t, const cf32* __restrict__ iq, int64_t n_samples,
                                const FrameDesc* __restrict__ frames, const CfoSegment* __restrict__ segs,
                                int n_segs, const cf32* __restrict__ taps_in, FrameStat* __restrict__ stats,
                                cf32* __restrict__ eq_hard, cf32* __restrict__ eq_soft,
                                cf32* __restrict__ soft_syms, float* __restrict__ llr_out = nullptr, int rho0 = 0,
                                FrameDesc* __restrict__ frames_copy = nullptr)
{
    static_assert(DEMAP < 0 || (STAGED && !FREQ_IN), "the fused demapper belongs to the staged receive-path form");
    constexpr bool FUSED = DEMAP >= 0;
    extern __shared__ unsigned char smem_raw[];
    const int N = t.fft_len, L = t.frame_length, tid = threadIdx.x, h = t.header_len;
    cf32* buf = reinterpret_cast<cf32*>(smem_raw);    // 4 x N : FFT scratch (four symbols per batch)
    Welford* wred = reinterpret_cast<Welford*>(buf + 4 * N); // blockDim/64 partials
    cf32* tw = reinterpret_cast<cf32*>(wred + (N + 63) / 64); // N per-pass twiddles
    [[maybe_unused]] cf32* st_soft = tw + N;          // STAGED: 4 x N data symbols of the running batch (lane k: column k)
    [[maybe_unused]] cf32* st_pil = FUSED ? tw + N : st_soft + 4 * N;  // STAGED: 4 x N pilot values of the lane's next four steps
    [[maybe_unused]] int16_t* s_occ = reinterpret_cast<int16_t*>(st_pil + 4 * N); // FUSED: shifted position of data carrier r
    [[maybe_unused]] cf32* park = reinterpret_cast<cf32*>(s_occ + ((N + 3) & ~3));  // FUSED: (rho0 + 1) x data carriers, serializer order
    [[maybe_unused]] float keep_re[kEqRegSlots], keep_im[kEqRegSlots];
    if constexpr (FUSED) {
#pragma unroll
        for (int e = 0; e < kEqRegSlots; e++) keep_re[e] = keep_im[e] = 0.0f;
    }
    [[maybe_unused]] const int n_lds = rho0 + 1;      // FUSED: destination symbols i < n_lds are parked in LDS
    tw[tid] = t.twiddle[tid];

    // (`frames` may be the host's pinned, coherent copy of the descriptors, read here once per workgroup; the kernels
    // behind this one read `frames_copy` in device memory)
    const FrameDesc fd = frames[blockIdx.x];
    if (frames_copy && threadIdx.x == 0) frames_copy[blockIdx.x] = fd;
    const int off = fd.carr_offset;
    const int total = L * N;
    const SegWindow sw = seg_window(segs, n_segs, fd.seg);

    // lane == destination carrier (shifted position k)
    const int k = tid;
    int lag = 0, ks = k + off;
    if (ks >= N) { lag = 1; ks -= N; }
    else if (ks < 0) { lag = -1; ks += N; }
    const int bin = fft_lds_swizzle((ks + N / 2) & (N - 1)); // natural-order FFT bin of the source column, as stored
    cf32 Hk = taps_in[size_t(fd.cand) * N + k];
    const float alpha = t.eq_alpha;
    const bool occupied = t.occupied[k] != 0;
    // The decided point feeds the second division of a step, which feeds the next symbol's first: a table read on
    // that chain must be short.  This frame's constellation goes to LDS; whether carrier k is a pilot (per pilot
    // set) becomes a bit mask in a register, and the pilot value of a step is requested before the first division.
    __shared__ cf32 s_pts[16];
    [[maybe_unused]] __shared__ float s_np[2];        // FUSED: the frame's npwr and its reciprocal for the demapper
    if (tid < 16) s_pts[tid] = t.cnst_pts[fd.cnst * 16 + tid];
    const cf32* pts = s_pts;
    const bool pilot_bits_ok = t.n_pilot_sets <= 32;
    unsigned pilot_bits = 0;
    if (pilot_bits_ok)
        for (int ps = 0; ps < t.n_pilot_sets; ps++) pilot_bits |= (t.pilot_mask[ps * N + k] ? 1u : 0u) << ps;
    Welford w = { 0.0, 0.0, 0.0 };
    const double* __restrict__ recip = t.recip_count;
    // A step must not wait for global memory: what it needs from the configuration tables is either constant per
    // lane (kept in registers: serializer rank / length for the usual single carrier set), small (the reciprocals of
    // a lane's own sample count, at most L + 1 of them: LDS) or known one step ahead (the pilot value, and rank /
    // length when the carrier sets rotate: requested during the previous step).  The three set indices advance by
    // one per symbol and wrap: counters, not modulo operations.
    // (the fused form serves frames of at most kEqFusedMaxSyms symbols and spends its LDS on parked data symbols)
    constexpr int kRecipHere = FUSED ? kEqFusedMaxSyms + 4 : kEqRecipLds;
    __shared__ double s_recip[kRecipHere];
    const bool recip_in_lds = L + 2 <= kRecipHere;
    if (recip_in_lds)
        for (int e = tid; e < L + 2; e += blockDim.x) s_recip[e] = recip[e];
    if constexpr (FUSED) {                            // one data-carrier set (STAGED): ofdm_serializer_vcc's order
        const int set = 1 % t.n_occ_sets, n_occ = t.occ_len[set];
        for (int e = tid; e < n_occ; e += blockDim.x) s_occ[e] = t.occ_idx[set * t.max_occ + e];
    }
    int pilot_carr_set = t.n_pilot_sets ? h % t.n_pilot_sets : 0;
    int pset = h % t.n_pilot_sym_sets;                // pilot symbol set of destination symbol 0
    int oset = 1 % t.n_occ_sets;                      // ofdm_serializer_vcc: symbols_skipped = 1
    const bool one_occ_set = t.n_occ_sets == 1;
    int occ_r = t.occ_rank[oset * N + k], occ_n = t.occ_len[oset];   // of the step about to run
    auto pilot_here = [&](int set) -> bool {
        return t.n_pilot_sets && (pilot_bits_ok ? ((pilot_bits >> set) & 1u) != 0 : t.pilot_mask[set * N + k] != 0);
    };
    cf32 pilot_next = { 0.0f, 0.0f };                 // pilot value of the step about to run
    if (!STAGED && pilot_here(pilot_carr_set)) pilot_next = t.pilot_val[pset * N + k];
    cf32* hard_row = eq_hard ? eq_hard + size_t(blockIdx.x) * total : nullptr;
    cf32* soft_row = eq_soft ? eq_soft + size_t(blockIdx.x) * total : nullptr;
    cf32* soft_out = soft_syms + size_t(blockIdx.x) * t.frame_capacity;
    int q_base = 0;                                   // data symbols in front of destination symbol i (serializer order)
    [[maybe_unused]] int staged = 0, staged_base = 0; // STAGED: steps run since stage_begin, q_base at stage_begin

    // frame_equalize for destination symbol i of this carrier (ofdm_adaptive_equalizer.cc:217-268)
    auto eq_step = [&](int i, cf32 y) -> cf32 {
        const bool is_pilot = STAGED ? (t.n_pilot_sets && ((pilot_bits >> pilot_carr_set) & 1u) != 0) : pilot_here(pilot_carr_set);
        cf32 hard_v = y, soft_v = { 0.0f, 0.0f };
        // pilots and data carriers run the same arithmetic -- equalise with the current tap, then pull the
        // tap towards y / (known or decided symbol) -- so the two cases share one pair of complex divisions
        // (lanes of one wave hold both kinds; separate branches would execute both pairs)
        if (!ablated(3) && (is_pilot || occupied)) {
            const cf32 eqv = cdiv(y, Hk);
            cf32 ref;
            if (is_pilot) {
                if (STAGED) ref = st_pil[staged * N + k];
                else ref = pilot_next;                // requested at the end of the previous step
                // mpsk_snr_est_simple::update on |pilot_eq| (Welford, double)
                const double x = double(sqrtf(eqv.re * eqv.re + eqv.im * eqv.im));
                w.n += 1.0;
                const double d = x - w.mean;
                // the table is in LDS (a pointer chosen at run time between LDS and global memory would become a flat
                // load, whose wait also drains the loads requested for the next step); frames too long for it divide
                double rc;
                if (recip_in_lds) rc = s_recip[int(w.n)];
                else rc = 1.0 / w.n;
                w.mean += dm_div_by_count(d, w.n, rc);
                w.m2 += d * (x - w.mean);
            } else {
                unsigned idx;
                switch (fd.cnst) { // decision_maker of each GR constellation
                case 1: idx = eqv.re > 0; break;
                case 2: idx = 2 * (eqv.im > 0) + (eqv.re > 0); break;
                case 3: idx = (fabsf(eqv.re) <= fabsf(eqv.im) ? 4u : 0u) | (eqv.re <= 0 ? 1u : 0u) | (eqv.im <= 0 ? 2u : 0u); break;
                default: {
                    const float level = sqrtf(0.1f);
                    idx = (eqv.im > 0 ? 8u : 0u) | (eqv.re < 0 ? 4u : 0u) | (fabsf(eqv.re) > 2 * level ? 2u : 0u) | (fabsf(eqv.im) > 2 * level ? 1u : 0u);
                } break;
                }
                ref = pts[idx];
                soft_v = eqv;
            }
            Hk = cadd(cscale(alpha, Hk), cdiv(cscale(1 - alpha, y), ref));
            hard_v = ref;
        }
        if (STAGED) {
            if (!FUSED) st_soft[staged * N + k] = soft_v;         // stored by stage_end
            else if (occ_r >= 0 && i < n_lds) park[i * occ_n + occ_r] = soft_v; // (later symbols: kept in registers by the caller)
            staged++;
            q_base += occ_n;
            pset = pset + 1 == t.n_pilot_sym_sets ? 0 : pset + 1;
            pilot_carr_set = t.n_pilot_sets ? (pilot_carr_set + 1 == t.n_pilot_sets ? 0 : pilot_carr_set + 1) : 0;
            return soft_v;
        }
        if (hard_row) hard_row[i * N + k] = hard_v;
        if (soft_row) soft_row[i * N + k] = soft_v;
        // ofdm_serializer_vcc (symbols_skipped = 1): data carrier r of OFDM symbol i is symbol q_base + r
        if (occ_r >= 0 && q_base + occ_r < fd.n_syms) soft_out[q_base + occ_r] = soft_v;
        q_base += occ_n;
        // next step's table entries: requested last (behind this step's stores: on this hardware loads and stores
        // complete in one queue, so whoever waits for a load also waits for every store issued before it) and first
        // looked at after the next step's first division
        pset = pset + 1 == t.n_pilot_sym_sets ? 0 : pset + 1;
        pilot_carr_set = t.n_pilot_sets ? (pilot_carr_set + 1 == t.n_pilot_sets ? 0 : pilot_carr_set + 1) : 0;
        oset = oset + 1 == t.n_occ_sets ? 0 : oset + 1;
        if (i + 1 < L) {
            if (pilot_here(pilot_carr_set)) pilot_next = t.pilot_val[pset * N + k];
            if (!one_occ_set) { occ_r = t.occ_rank[oset * N + k]; occ_n = t.occ_len[oset]; }
        }
        return soft_v;
    };

    // STAGED: pilot values of the lane's next four steps (loads without a branch between them: one wait for all)
    auto stage_begin = [&]() {
        if (!STAGED) return;
        staged = 0;
        staged_base = q_base;
        if (pilot_bits) {
            int ps = pset;
            cf32 pv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                pv[u] = t.pilot_val[ps * N + k];
                ps = ps + 1 == t.n_pilot_sym_sets ? 0 : ps + 1;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) st_pil[u * N + k] = pv[u];
        }
    };
    // STAGED: the data symbols of the steps just run, in serializer order (symbols_skipped = 1: data carrier r of the
    // u-th of these steps is symbol staged_base + u occ_n + r)
    auto stage_end = [&]() {
        if (!STAGED || FUSED) return;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = staged_base + u * occ_n + occ_r;
            if (u < staged && occ_r >= 0 && idx < fd.n_syms) soft_out[idx] = st_soft[u * N + k];
        }
    };

    __syncthreads();                                  // s_pts, tw
    [[maybe_unused]] cf32 post_v = { 0.0f, 0.0f };    // FUSED: the symbol equalised from the zero row behind the frame (lag > 0)
    // (the zero fill of the shift passes through the symbol's phase correction like every other position -- the
    // reference clears the output, copies the shifted frame into it and then multiplies ALL of it -- which decides the
    // signs of those zeros)
    auto zero_at = [&](int i) -> cf32 {
        cf32 pc = { 1.0f, -0.0f };
        if (off != 0) pc = expj(float(-(2.0 * kPi) * off * t.cp_len / N * (i + 1)));
        return cmul(cf32{ 0.0f, 0.0f }, pc);
    };
    if (lag < 0) { // its source would be row -1: the zero fill of the shift
        stage_begin();
        eq_step(0, zero_at(0));                      // FUSED: destination 0 is always among the parked rows
        stage_end();
    }
    for (int i0 = 0; i0 < L; i0 += 4) {
        const int nb = min(4, L - i0);
        if constexpr (FREQ_IN) {
            const int nat = fft_lds_swizzle((tid + N / 2) & (N - 1));   // where shifted position tid sits in a row
            for (int u = 0; u < nb; u++) buf[u * N + nat] = iq[(size_t(blockIdx.x) * L + (i0 + u)) * N + tid];
            __syncthreads();
        } else {
            load_and_fft4<kAblate>(t, iq, n_samples, fd.trigger, 2 + h + i0, nb, segs, n_segs, sw, buf, tw);
        }
        stage_begin();
        if constexpr (FUSED) {
            // the four steps written out: their results go to the register slots of source rows i0 .. i0 + 3
            cf32 sv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                sv[u] = cf32{ 0.0f, 0.0f };
                const int i = i0 + u - lag;
                if (u < nb && i >= 0 && i < L) {
                    const cf32 v = buf[u * N + bin];
                    cf32 pc = { 1.0f, -0.0f };
                    if (off != 0) pc = expj(float(-(2.0 * kPi) * off * t.cp_len / N * (i + 1)));
                    sv[u] = eq_step(i, cmul(v, pc));
                }
            }
            if (i0 >= rho0) keep_window<true, 0, 4>((i0 - rho0) >> 2, keep_re, keep_im, sv);
        } else
        for (int u = 0; u < nb; u++) {
            const int i = i0 + u - lag;               // destination symbol fed by source row i0 + u
            if (i >= 0 && i < L) {
                const cf32 v = buf[u * N + bin];
                // out *= expj(-2 pi off cp / N (sym + 1)) with sym of the destination
                cf32 pc = { 1.0f, -0.0f };            // expj(-0.0f): the value the formula takes for off == 0
                if (off != 0) pc = expj(float(-(2.0 * kPi) * off * t.cp_len / N * (i + 1)));
                eq_step(i, cmul(v, pc));
            }
        }
        stage_end();
        __syncthreads();                              // the next batch overwrites the scratch rows
    }
    if (lag > 0) { stage_begin(); post_v = eq_step(L - 1, zero_at(L - 1)); stage_end(); } // its source would be row L

    // combine the per-carrier Welford states (Chan); double, fixed tree order.  The reciprocals of the merged counts
    // (at most L x pilots-per-set of them) go to the FFT scratch first, which is free now.
    double* mrec = reinterpret_cast<double*>(buf);
    const int n_mrec = min(L * t.n_pilots + 2, 4 * N);
    {
        double rv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) rv[j] = recip[min(tid + j * int(blockDim.x), n_mrec - 1)];
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (tid + j * int(blockDim.x) < n_mrec) mrec[tid + j * int(blockDim.x)] = rv[j];
    }
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
        Welford other = { __shfl_down(w.n, o), __shfl_down(w.mean, o), __shfl_down(w.m2, o) };
        w = welford_merge(w, other, recip, mrec, n_mrec);
    }
    if ((tid & 63) == 0) wred[tid >> 6] = w;
    __syncthreads();
    if (tid == 0) {
        Welford a = wred[0];
        for (int i = 1; i < int(blockDim.x + 63) / 64; i++) a = welford_merge(a, wred[i], recip, mrec, n_mrec);
        // mpsk_snr_est_simple::snr()/noise(); get_noise() (equalizer.cc:279-287)
        const double signal_db = 10.0 * dm_log10_64(fabs(a.mean * a.mean));
        const double noise_db = 10.0 * dm_log10_64(2.0 * a.m2 / a.n);
        const double noise_lin = dm_pow10_64(noise_db / 10.0);
        stats[blockIdx.x] = { signal_db - noise_db, noise_db, noise_lin };
        if constexpr (FUSED) {
            // sigma = sqrt(noise) passed as npwr (soft_cf_impl.cc:101,144) and its correctly rounded single reciprocal
            const float npwr = float(sqrt(noise_lin));
            s_np[0] = npwr;
            s_np[1] = float(1.0 / double(npwr));
        }
    }
    if constexpr (FUSED) {
        // ---- ofdm_serializer_vcc + ofdm_adaptive_constellation_soft_cf on the symbols this workgroup kept
        __syncthreads();                              // s_np, parked rows; the reciprocal table in the scratch rows is dead
        const float npwr = s_np[0], rinv = s_np[1];
        const int n_syms = fd.n_syms;                 // payload == 0: frame dropped after the equaliser (:218,254)
        const int bps = fd.cnst == 4 ? 4 : fd.cnst;
        const int n_occ = occ_n;                      // data carriers per OFDM symbol (one set)
        const uint32_t base = fd.llr_off;
        float* const llr_frame = llr_out + size_t(base);
        cf32* const soft_out_f = soft_syms ? soft_syms + size_t(blockIdx.x) * t.frame_capacity : nullptr;
        auto demap_one = [&](int q, cf32 sy) {
            if (soft_out_f) soft_out_f[q] = sy;
            float* dst = llr_frame + size_t(q) * bps;
            float o[4];
            switch (bps) {
            case 1: soft_demap_symbol<1, DEMAP>(sy, pts, npwr, rinv, o); dst[0] = o[0]; break;
            case 2:
                soft_demap_symbol<2, DEMAP>(sy, pts, npwr, rinv, o);
                if ((base & 1u) == 0) *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
                else { dst[0] = o[0]; dst[1] = o[1]; }
                break;
            case 3: soft_demap_symbol<3, DEMAP>(sy, pts, npwr, rinv, o); dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; break;
            default:
                soft_demap_symbol<4, DEMAP>(sy, pts, npwr, rinv, o);
                if ((base & 3u) == 0) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                else { dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3]; }
                break;
            }
        };
        // the parked rows: already in serializer order
        const int n_parked = min(n_syms, min(n_lds, L) * n_occ);
        for (int q = tid; q < n_parked; q += N) demap_one(q, park[q]);
        // the rows kept in registers, four OFDM symbols at a time through the scratch rows: destination symbol
        // rho0 + 4 B + u of this carrier sits in slot 4 B + u + lag - 1 (slot kEqRegSlots: the zero row behind the frame)
        for (int B = 0; rho0 + 4 * B < L && (rho0 + 4 * B + 4 > n_lds) && (rho0 + 4 * B) * n_occ < n_syms; B++) {
            const int i_first = rho0 + 4 * B, nb = min(4, L - i_first);
            cf32 w6[6];
            keep_window<false, -1, 6>(B, keep_re, keep_im, w6);
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int u = j - lag - 1;
                if (u >= 0 && u < nb && i_first + u >= n_lds) {
                    cf32 v = w6[j];
                    if (4 * B - 1 + j == L - rho0) v = post_v;   // the slot behind the last source row
                    buf[u * N + k] = v;
                }
            }
            __syncthreads();
            const int n_here = nb * n_occ;
            for (int qq = tid; qq < n_here; qq += N) {
                const int u = (qq >= n_occ ? 1 : 0) + (qq >= 2 * n_occ ? 1 : 0) + (qq >= 3 * n_occ ? 1 : 0);
                const int q = i_first * n_occ + qq;   // index in the frame's symbol stream
                if (i_first + u >= n_lds && q < n_syms) demap_one(q, buf[u * N + s_occ[qq - u * n_occ]]);
            }
            __syncthreads();                          // the next batch overwrites the rows
        }
    }
}

// ofdm_adaptive_constellation_soft_cf over all frames of a batch: thread == data symbol (blockIdx.x == frame)
template <int MODE>
__global__ __launch_bounds__(256) void demap_kernel(RxTables t, const FrameDesc* __restrict__ frames,
                                                   const FrameStat* __restrict__ stats,
                                                   const cf32* __restrict__ soft_syms, float* __restrict__ llr_out)
{
    const int f = blockIdx.x;
    const int n_syms = frames[f].n_syms;              // payload == 0: frame dropped after the equaliser (:218,254)
    const int q = blockIdx.y * blockDim.x + threadIdx.x;
    if (q >= n_syms) return;
    const int cnst = frames[f].cnst;
    const int bps = cnst == 4 ? 4 : cnst;
    const cf32* pts = t.cnst_pts + cnst * 16;
    const float npwr = float(sqrt(stats[f].noise_lin)); // sigma = sqrt(noise) passed as npwr (:101,144)
    const float rinv = float(1.0 / double(npwr));        // the correctly rounded single reciprocal
    const cf32 s = ablated(34) ? cf32{ 0.3f + 1e-3f * float(q & 63), -0.9f + 1e-3f * float(threadIdx.x & 31) }
                               : soft_syms[size_t(f) * t.frame_capacity + q];
    const uint32_t base = frames[f].llr_off;
    float* dst = llr_out + size_t(base) + size_t(q) * bps;
    float o[4];
    const bool ord = __all(demap_operands_ordinary(s, npwr));   // wave-uniform: see soft_demap_symbol
    switch (bps) {
    case 1: soft_demap_symbol<1, MODE>(s, pts, npwr, rinv, o); dst[0] = o[0]; break;
    case 2:
        if (ord) soft_demap_symbol<2, MODE, true>(s, pts, npwr, rinv, o);
        else soft_demap_symbol<2, MODE>(s, pts, npwr, rinv, o);
        // a symbol's soft bits leave in ONE store when the frame's position in the stream allows it (a frame starts
        // wherever the previous ones ended): four dword stores at a 16-byte stride make the memory pipeline handle
        // 64 four-byte pieces each
        if ((base & 1u) == 0) *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
        else { dst[0] = o[0]; dst[1] = o[1]; }
        break;
    case 3:
        if (ord) soft_demap_symbol<3, MODE, true>(s, pts, npwr, rinv, o);
        else soft_demap_symbol<3, MODE>(s, pts, npwr, rinv, o);
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
        break;
    default:
        if (ord) soft_demap_symbol<4, MODE, true>(s, pts, npwr, rinv, o);
        else soft_demap_symbol<4, MODE>(s, pts, npwr, rinv, o);
        if (ablated(33) && o[0] + o[1] + o[2] + o[3] != 12345.678f) break;
        if ((base & 3u) == 0) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else { dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3]; }
        break;
    }
}

// no-FEC tail: ofdm_adaptive_constellation_decoder_cb + ofdm_adaptive_frame_pack_bb; the soft symbols are
// sliced with the decision makers of the GR constellations, repacked bps -> 8 LSB first, CRC-32 checked.
__global__ __launch_bounds__(256) void nofec_tail_kernel(RxTables t, const FrameDesc* __restrict__ frames,
                                                        const cf32* __restrict__ soft_syms,
                                                        uint8_t* __restrict__ nofec_bytes, int32_t* __restrict__ nofec_info,
                                                        int nofec_stride)
{
    const FrameDesc fd = frames[blockIdx.x];
    if (fd.n_syms <= 0) return;
    const int tid = threadIdx.x;
    const int bps = fd.cnst == 4 ? 4 : fd.cnst;
    const cf32* soft = soft_syms + size_t(blockIdx.x) * t.frame_capacity;
    uint8_t* bytes = nofec_bytes + size_t(blockIdx.x) * nofec_stride;
    const int n_bits = fd.n_syms * bps;
    int n_written = n_bits / 8 + ((n_bits % 8) ? 1 : 0);
    if (n_bits % 8) --n_written;
    for (int b = tid; b < n_written; b += blockDim.x) {
        unsigned v = 0;
        for (int j = 0; j < 8; j++) {
            const int bit = b * 8 + j, q = bit / bps, jb = bit % bps;
            const cf32 s = soft[q];
            unsigned idx;
            switch (fd.cnst) {
            case 1: idx = s.re > 0; break;
            case 2: idx = 2 * (s.im > 0) + (s.re > 0); break;
            case 3: idx = (fabsf(s.re) <= fabsf(s.im) ? 4u : 0u) | (s.re <= 0 ? 1u : 0u) | (s.im <= 0 ? 2u : 0u); break;
            default: { const float level = sqrtf(0.1f);
                idx = (s.im > 0 ? 8u : 0u) | (s.re < 0 ? 4u : 0u) | (fabsf(s.re) > 2 * level ? 2u : 0u) | (fabsf(s.im) > 2 * level ? 1u : 0u); } break;
            }
            v |= ((idx >> jb) & 1u) << j;
        }
        bytes[b] = uint8_t(v);
    }
    __syncthreads();
    if (tid == 0) {
        int ok = 0;
        if (n_written >= 4) {
            const uint32_t crc = crc_bits(bytes, n_written - 4, 32, 0x04C11DB7u, 0xFFFFFFFFu, 0xFFFFFFFFu, true, true);
            ok = 1;
            for (int i = 0; i < 4; i++) if (bytes[n_written - 4 + i] != ((crc >> (8 * i)) & 0xFF)) ok = 0;
        }
        nofec_info[2 * blockIdx.x] = ok;
        nofec_info[2 * blockIdx.x + 1] = n_written;
    }
}

// ------------------------------------------------------------------------------------------
// K11: per transport block -- repack decoded bits to bytes (LSB first), CRC-32 verify
// (ofdm_adaptive_fec_decoder_impl.cc:161-181), average iteration count (tb_decoder.cc:191).
// ------------------------------------------------------------------------------------------
// CRC-32 register after appending k zero bytes: the "append 2^j zero bytes" maps are linear over GF(2) and
// given as 32 columns each (zmat[j * 32 + b] = image of bit b); binary expansion of k.
__device__ __forceinline__ uint32_t crc32_shift(uint32_t v, unsigned k, const uint32_t* __restrict__ zmat)
{
    for (int j = 0; k; j++, k >>= 1) {
        if (k & 1u) {
            const uint32_t* m = zmat + j * 32;
            uint32_t acc = 0;
#pragma unroll
            for (int b = 0; b < 32; b++) acc ^= ((v >> b) & 1u) ? m[b] : 0u;
            v = acc;
        }
    }
    return v;
}

// PACKED: `bits` holds the decoder's packed output (1 bit per bit, LSB first; block at bit offset bits_off, a multiple
// of 8) -- the block's bytes are read as they are; otherwise one byte per bit, repacked here.
template <bool PACKED>
__global__ __launch_bounds__(64) void tb_check_kernel(const TbGather* __restrict__ tbs, int n_tbs,
                                                     const uint8_t* __restrict__ bits,
                                                     const int32_t* __restrict__ nit,
                                                     uint8_t* __restrict__ staging, int stride,
                                                     TbResult* __restrict__ res, const uint32_t* __restrict__ zmat_g)
{
    // CRC-32 (reflected polynomial 0xEDB88320 == the reference's crc(32, 0x04C11DB7, ~0, ~0, refin, refout)) of
    // the TB's bytes, computed by all 64 lanes: the register update is linear, so every lane runs the byte-wise
    // table recurrence over its own slice from a zero register, moves the result past the bytes that follow its
    // slice with the zero-append maps, and the XOR of all lanes plus the moved initial value is the CRC.
    // LDS: 256-entry table, 16 x 32 map columns, the TB's packed bytes.
    extern __shared__ uint32_t tb_smem[];
    uint32_t* T = tb_smem;
    uint32_t* zmat = tb_smem + 256;
    uint8_t* bytes_l = reinterpret_cast<uint8_t*>(tb_smem + 256 + 16 * 32);
    const int tb = blockIdx.x;
    if (tb >= n_tbs) return;
    const int lane = threadIdx.x;
    {   // the zero-append maps and, behind them, the byte table (both built once on the host: a workgroup that formed its own
        // 256 entries spent a fifth of its instructions on them); loads in one group without a branch between them: one
        // memory latency for all twelve
        uint32_t z[12];
#pragma unroll
        for (int j = 0; j < 12; j++) z[j] = zmat_g[lane + 64 * j];
#pragma unroll
        for (int j = 0; j < 8; j++) zmat[lane + 64 * j] = z[j];
#pragma unroll
        for (int j = 0; j < 4; j++) T[lane + 64 * j] = z[8 + j];
    }
    // the block's place in the decoder's output comes from its gather recipe (one table for expand_cw and this kernel)
    const TbDesc d = { tbs[tb].bits_off, uint32_t(tbs[tb].tb_payload), tbs[tb].cw_first, uint32_t(tbs[tb].n_cw) };
    uint8_t* bytes = staging + size_t(tb) * stride;
    const int n_bytes = int(d.n_bits / 8);
    if constexpr (PACKED) {
        const uint8_t* src = bits + size_t(d.bits_off / 8);
        for (int b0 = lane; b0 < n_bytes; b0 += 64 * 4) {
            uint8_t v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = src[min(b0 + 64 * j, n_bytes - 1)];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int b = b0 + 64 * j;
                if (b < n_bytes) { bytes_l[b] = v[j]; bytes[b] = v[j]; }
            }
        }
    } else
    for (int b0 = lane; b0 < n_bytes; b0 += 64 * 4) {
        unsigned long long w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int b = min(b0 + 64 * j, n_bytes - 1);
            w[j] = *reinterpret_cast<const unsigned long long*>(bits + size_t(d.bits_off) + size_t(b) * 8);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int b = b0 + 64 * j;
            if (b < n_bytes) {
                // repack(1,8) LSB first: bit j of the byte is the LSB of input byte j
                const unsigned v = unsigned(((w[j] & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
                bytes_l[b] = uint8_t(v);
                bytes[b] = uint8_t(v);
            }
        }
    }
    int it_sum = 0;
    for (unsigned c = lane; c < d.n_cw; c += 64) it_sum += nit[d.cw_first + c];
    for (int o = 32; o > 0; o >>= 1) it_sum += __shfl_down(it_sum, o);
    __syncthreads();
    int ok = 0;
    if (n_bytes >= 4) {
        const int len = n_bytes - 4;
        const int per = (len + 63) / 64;
        const int lo = min(lane * per, len), hi = min(lo + per, len);
        uint32_t r = 0;
        for (int i = lo; i < hi; i++) r = T[(r ^ bytes_l[i]) & 0xFF] ^ (r >> 8);
        r = crc32_shift(r, unsigned(len - hi), zmat);
        if (lane == 0) r ^= crc32_shift(0xFFFFFFFFu, unsigned(len), zmat);
        for (int o = 32; o > 0; o >>= 1) r ^= __shfl_xor(r, o);
        const uint32_t crc = ~r;
        ok = 1;
        for (int k = 0; k < 4; k++) if (bytes_l[len + k] != ((crc >> (8 * k)) & 0xFF)) ok = 0;
    }
    if (lane == 0) {
        TbResult r{};
        r.user_bits = int(d.n_bits) - 32;
        r.avg_it = d.n_cw ? it_sum / int(d.n_cw) : 0;
        r.crc_ok = ok;
        res[tb] = r;
    }
}


// ------------------------------------------------------------------------------------------
// Stand-alone stages for block shims and unit tests (the fused kernels above are what the receive path
// runs; these expose the same device arithmetic one block of the reference graph at a time).
// ------------------------------------------------------------------------------------------
struct CnstPoints {
    cf32 p[16];
};

// ofdm_adaptive_constellation_soft_cf (soft_cf_impl.cc:89-148): bps soft bits per symbol, LSB first
template <int MODE>
__global__ __launch_bounds__(256) void soft_demap_kernel(const cf32* __restrict__ syms, int64_t n, int bps, CnstPoints pts,
                                                        float npwr, float rinv, float* __restrict__ llr)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* o = llr + i * bps;
    const cf32 s = syms[i];
    switch (bps) {
    case 1: soft_demap_symbol<1, MODE>(s, pts.p, npwr, rinv, o); break;
    case 2: soft_demap_symbol<2, MODE>(s, pts.p, npwr, rinv, o); break;
    case 3: soft_demap_symbol<3, MODE>(s, pts.p, npwr, rinv, o); break;
    default: soft_demap_symbol<4, MODE>(s, pts.p, npwr, rinv, o); break;
    }
}

// GR fft_vcc(fft_len, forward, window = (), shift = true): one workgroup of N/2 lanes per vector, the radix-2
// butterflies and twiddle products of the receive / transmit kernels
__global__ void fft_batch_kernel(const cf32* __restrict__ in, cf32* __restrict__ out, int N, int forward,
                                 const cf32* __restrict__ tw_g, const uint16_t* __restrict__ br_g)
{
    extern __shared__ unsigned char smem_raw[];
    cf32* x = reinterpret_cast<cf32*>(smem_raw);
    cf32* tw = x + N;
    const int tid = threadIdx.x, hN = N / 2;
    const cf32* src = in + size_t(blockIdx.x) * N;
    cf32* dst = out + size_t(blockIdx.x) * N;
    { // forward twiddles are exp(-j 2 pi k / N); the inverse transform uses their conjugates
        cf32 w = tw_g[tid];
        if (!forward) w.im = -w.im;
        tw[tid] = w;
    }
    for (int k = tid; k < N; k += hN) {
        const int from = forward ? k : ((k + hN) & (N - 1)); // inverse: swap halves first
        x[br_g[k]] = src[from];
    }
    __syncthreads();
    const int logN = __ffs(N) - 1;
    for (int st = 0; st < logN; st++) {
        const int half = 1 << st;
        const int grp = tid >> st, k = tid & (half - 1);
        const int a = (grp << (st + 1)) + k, b = a + half;
        const cf32 w = tw[k << (logN - st - 1)];
        const cf32 u = x[a];
        const cf32 vv = cmul_tw(x[b], w);
        x[a] = cadd(u, vv);
        x[b] = csub(u, vv);
        __syncthreads();
    }
    for (int k = tid; k < N; k += hN) dst[forward ? ((k + hN) & (N - 1)) : k] = x[k]; // forward: swap halves last
}

// CRC-32 of one byte string (crc_util.cc:41-61 / the CRC of ofdm_adaptive_frame_pack_bb), all 64 lanes
__global__ __launch_bounds__(64) void crc32_kernel(const uint8_t* __restrict__ bytes, int64_t len,
                                                  const uint32_t* __restrict__ zmat_g, uint32_t* __restrict__ out)
{
    __shared__ uint32_t T[256];
    __shared__ uint32_t zmat[32 * 32];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) {
        uint32_t c = uint32_t(i);
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        T[i] = c;
    }
    for (int i = lane; i < 32 * 32; i += 64) zmat[i] = zmat_g[i];
    __syncthreads();
    const int64_t per = (len + 63) / 64;
    const int64_t lo = min(int64_t(lane) * per, len), hi = min(lo + per, len);
    uint32_t r = 0;
    for (int64_t i = lo; i < hi; i++) r = T[(r ^ bytes[i]) & 0xFF] ^ (r >> 8);
    r = crc32_shift(r, unsigned(len - hi), zmat);
    if (lane == 0) r ^= crc32_shift(0xFFFFFFFFu, unsigned(len), zmat);
    for (int o = 32; o > 0; o >>= 1) r ^= __shfl_xor(r, o);
    if (lane == 0) *out = ~r;
}

// K9: transport-block reassembly as index arithmetic -- one CwDesc per codeword from the TB recipe.
// Codeword i carries k_i = floor(P/n) + (i < P mod n) payload bits (the iterated ceil split of
// tb_decoder.cc:168-172) and starts at receive-buffer position i*ncheck + sum_{j<i} k_j.
__global__ __launch_bounds__(64) void expand_cw_kernel(const TbGather* __restrict__ tbs, int n_tbs,
                                                      CwDesc* __restrict__ out)
{
    const int tb = blockIdx.x;
    if (tb >= n_tbs) return;
    const TbGather g = tbs[tb];
    const int q = g.n_cw > 0 ? g.tb_payload / g.n_cw : 0, r = g.n_cw > 0 ? g.tb_payload % g.n_cw : 0;
    for (int i = threadIdx.x; i < g.n_cw; i += 64) {
        const int k_i = q + (i < r ? 1 : 0);
        const int before = i * q + min(i, r);
        const int k_used = min(max(k_i, 0), int(g.k));
        const long want = g.ncheck + k_used;
        long rcv_idx = long(i) * g.ncheck + before;
        CwDesc d;
        d.out_off = g.bits_off + uint32_t(before);
        d.k_used = uint16_t(k_used);
        d.len1 = 0; d.pad = 0; d.llr_off1 = 0;
        int si = 0;
        long seg_start = 0;
        while (si < g.n_seg && seg_start + g.seg_len[si] <= rcv_idx) { seg_start += g.seg_len[si]; si++; }
        if (si >= g.n_seg || g.seg_off[si] == 0xFFFFFFFFu) {
            d.llr_off0 = 0; d.len0 = 0; // nothing (or a gap) at the codeword's start: never received
        } else {
            const long in_seg = rcv_idx - seg_start;
            const long avail0 = g.seg_len[si] - in_seg;
            d.llr_off0 = g.seg_off[si] + uint32_t(in_seg);
            if (avail0 >= want) d.len0 = uint16_t(want);
            else {
                d.len0 = uint16_t(avail0);
                if (si + 1 < g.n_seg && g.seg_off[si + 1] != 0xFFFFFFFFu) {
                    d.llr_off1 = g.seg_off[si + 1];
                    d.len1 = uint16_t(min(long(g.seg_len[si + 1]), want - avail0));
                }
            }
        }
        out[g.cw_first + i] = d;
    }
}

// Exclusive scan of the per-block output lengths (single workgroup; block counts are O(frames)).  Every thread owns a run
// of consecutive entries, a multiple of four, read and written as 16-byte vectors with the loads of a run issued
// together (one entry per load, each waited for in turn, was 126 serial memory round trips per thread: 0.095 ms for
// 64 k blocks).
__global__ __launch_bounds__(1024) void scan_lengths_kernel(const int32_t* __restrict__ len, int n,
                                                           uint32_t* __restrict__ offs,
                                                           uint32_t* __restrict__ total)
{
    __shared__ uint32_t part[1024];
    const int tid = threadIdx.x;
    const int per = ((n + 1023) / 1024 + 3) & ~3;
    const int lo = min(n, tid * per), hi = min(n, lo + per);
    const bool vec = (reinterpret_cast<uintptr_t>(len) & 15) == 0 && (reinterpret_cast<uintptr_t>(offs) & 15) == 0;
    uint32_t s = 0;
    int i = lo;
    if (vec) {
#pragma unroll 4
        for (; i + 4 <= hi; i += 4) {
            const int4 v = *reinterpret_cast<const int4*>(len + i);
            s += uint32_t(v.x) + uint32_t(v.y) + uint32_t(v.z) + uint32_t(v.w);
        }
    }
    for (; i < hi; i++) s += uint32_t(len[i]);
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        uint32_t v = (tid >= o) ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    i = lo;
    if (vec) {
#pragma unroll 4
        for (; i + 4 <= hi; i += 4) {
            const int4 v = *reinterpret_cast<const int4*>(len + i);
            uint4 o;
            o.x = run; run += uint32_t(v.x);
            o.y = run; run += uint32_t(v.y);
            o.z = run; run += uint32_t(v.z);
            o.w = run; run += uint32_t(v.w);
            *reinterpret_cast<uint4*>(offs + i) = o;
        }
    }
    for (; i < hi; i++) { offs[i] = run; run += uint32_t(len[i]); }
    if (tid == 1023) *total = part[1023];
}

__global__ void compact_kernel(const uint8_t* __restrict__ staging, int stride, const int32_t* __restrict__ len,
                               const uint32_t* __restrict__ offs, int n, uint8_t* __restrict__ out)
{
    const int b = blockIdx.x;
    if (b >= n) return;
    const int l = len[b];
    const uint8_t* src = staging + size_t(b) * stride;
    uint8_t* dst = out + offs[b];
    for (int i = threadIdx.x; i < l; i += blockDim.x) dst[i] = src[i];
}

// per-block output length: user bytes of a CRC-ok TB (FEC) or n_written-4 of every frame (no FEC)
__global__ void tb_lengths_kernel(const TbResult* __restrict__ res, int n, int32_t* __restrict__ len)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) len[i] = res[i].crc_ok ? res[i].user_bits / 8 : 0;
}
__global__ void nofec_lengths_kernel(const int32_t* __restrict__ info, int n, int32_t* __restrict__ len)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) len[i] = info[2 * i + 1] >= 4 ? info[2 * i + 1] - 4 : 0;
}

// ------------------------------------------------------------------------------------------ launchers
int launch_sync_edges(const cf32* iq, int64_t n, int64_t m_begin, int N, float thr, unsigned long long* edges,
                      unsigned int* n_edges, unsigned int max_edges, float* metric_out, hipStream_t st,
                      unsigned long long* check)
{
    if (m_begin < 0 || m_begin > n) return DTL_ERR_INVALID_ARG;
    const int R = N <= 512 ? 8 : 16;
    if (N < 64 || (N % (2 * R)) != 0 || N / R > 64) return DTL_ERR_UNSUPPORTED; // history strips live in one wave
    const int threads = kSyncTile / R;
    const int n_raw = kSyncTile + N + N / 2;
    const size_t lds = size_t(N / R + threads) * 3 * sizeof(double) + size_t(n_raw + n_raw / 16 + 2) * sizeof(cf32);
    const int64_t grid = (n + kSyncTile - 1) / kSyncTile - m_begin / kSyncTile;
    if (grid <= 0) return DTL_OK;
    using Kern = void (*)(const cf32*, int64_t, int64_t, int, float, unsigned long long*, unsigned int*, unsigned int, float*,
                          unsigned long long*);
    // the single-precision pre-test serves the receive path (no metric output); DTL_SYNC_NO_PRETEST=1 keeps every strip on
    // the exact path (A/B timing, tests); `check`: both paths on every strip, disagreements counted
    static const bool no_pre = getenv("DTL_SYNC_NO_PRETEST") != nullptr;
    const int variant = metric_out ? 1 : (check ? 3 : (no_pre ? 0 : 2));
    const int which = variant + (R == 16 ? 4 : 0);
    static const Kern kerns[8] = { sync_edges_kernel<false, 8>, sync_edges_kernel<true, 8>, sync_edges_kernel<false, 8, true>,
                                   sync_edges_kernel<false, 8, true, true>,
                                   sync_edges_kernel<false, 16>, sync_edges_kernel<true, 16>, sync_edges_kernel<false, 16, true>,
                                   sync_edges_kernel<false, 16, true, true> };
    static size_t attr_lds[8] = { 64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024 };
    static std::mutex attr_mu;                            // (receivers may run in several host threads)
    std::lock_guard<std::mutex> attr_lock(attr_mu);
    if (lds > attr_lds[which]) {
        DTL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[which]),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        attr_lds[which] = lds;
    }
    hipLaunchKernelGGL(kerns[which], dim3((unsigned)grid), dim3(threads), lds, st, iq, n, m_begin, N, thr, edges, n_edges,
                       max_edges, metric_out, check);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

int launch_trigger_angle(const cf32* iq, int64_t n, int N, const int64_t* trig, int n_trig, float* angle,
                         hipStream_t st)
{
    if (n_trig <= 0) return DTL_OK;
    hipLaunchKernelGGL(trigger_angle_kernel, dim3(n_trig), dim3(64), 0, st, iq, n, N, trig, n_trig, angle);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

// participating carriers per round of the integer-offset search: the largest power of two (<= 64) whose product table
// (offsets x (chunk + 1) complex values) stays within 14 KB -- 1024-point headers: 16, 256-point: 64, 64-point: 32 (= all)
static int header_search_chunk(const RxTables& t)
{
    const int n_g = (t.g_max - t.g_min) / 2 + 1;
    int ch = 64;
    while (ch > 1 && (size_t(n_g) * (ch + 1) * sizeof(cf32) > 14 * 1024 || ch / 2 >= std::max(t.n_sync_act, 1))) ch /= 2;
    return ch;
}

size_t header_lds_bytes(const RxTables& t)
{
    const int n_g = (t.g_max - t.g_min) / 2 + 1;
    size_t b = size_t(2 + t.header_len + 1) * t.fft_len * sizeof(cf32) + size_t(t.fft_len) * sizeof(cf32) +
               size_t(t.fft_len / 2 + 2) * sizeof(float) + size_t((t.hdr_bits + 15) & ~15) +
               size_t(t.fft_len) * sizeof(cf32) + size_t(n_g) * (header_search_chunk(t) + 1) * sizeof(cf32) +
               size_t(t.n_sync_act) * (sizeof(cf32) + sizeof(uint16_t)) + 16;
    return (b + 15) & ~size_t(15);
}

static size_t frame_eq_lds_bytes(const RxTables& t, bool staged, bool fused, int rho0 = 0);

int launch_header(const RxTables& t, const cf32* iq, int64_t n, const Candidate* cands, int n_cand,
                  const CfoSegment* segs, int n_segs, HeaderRec* recs, cf32* taps, uint8_t* hdr_bits,
                  hipStream_t st)
{
    if (n_cand <= 0) return DTL_OK;
    const size_t lds = header_lds_bytes(t);
    static size_t attr_lds = 64 * 1024;
    static std::mutex attr_mu;
    std::lock_guard<std::mutex> attr_lock(attr_mu);
    if (lds > attr_lds) {
        DTL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(header_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        DTL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(header_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        attr_lds = lds;
    }
    hipLaunchKernelGGL(header_kernel<false>, dim3(n_cand), dim3(t.fft_len), lds, st, t, iq, n, cands,
                       segs, n_segs, recs, taps, hdr_bits, header_search_chunk(t), HeaderStageIo{});
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

// dtl_chanest / dtl_header_decode: n blocks of frequency-domain rows (see HeaderStageIo)
int launch_header_stage(const RxTables& t, int n, const HeaderStageIo& io, HeaderRec* recs, cf32* taps_out,
                        uint8_t* hdr_bits, hipStream_t st)
{
    if (n <= 0) return DTL_OK;
    const size_t lds = header_lds_bytes(t);
    if (lds > 64 * 1024) {
        DTL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(header_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    }
    hipLaunchKernelGGL(header_kernel<true>, dim3(n), dim3(t.fft_len), lds, st, t, nullptr, int64_t(0), nullptr, nullptr, 0,
                       recs, taps_out, hdr_bits, header_search_chunk(t), io);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

// dtl_equalize_frames: n frames of L x N frequency-domain vectors -> sliced / pre-slice symbols, serializer output,
// SNR estimate.  Always the general (not STAGED) form: the rows are requested by the caller.
int launch_frame_eq_stage(const RxTables& t, const cf32* freq, const FrameDesc* frames, int n_frames, const cf32* taps,
                          FrameStat* stats, cf32* eq_hard, cf32* eq_soft, cf32* soft_syms, hipStream_t st)
{
    if (n_frames <= 0) return DTL_OK;
    const size_t lds = frame_eq_lds_bytes(t, false, false);
    if (lds > 64 * 1024) { set_last_error("frame equaliser: FFT scratch does not fit LDS"); return DTL_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL((frame_eq_kernel<false, true>), dim3(n_frames), dim3(t.fft_len), lds, st, t, freq, int64_t(0), frames,
                       nullptr, 0, taps, stats, eq_hard, eq_soft, soft_syms, nullptr, 0);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

static size_t frame_eq_lds_bytes(const RxTables& t, bool staged, bool fused, int rho0)
{
    size_t b = size_t(4 * t.fft_len) * sizeof(cf32) + size_t((t.fft_len + 63) / 64) * sizeof(Welford) +
               size_t(t.fft_len) * sizeof(cf32);
    if (staged) b += size_t((fused ? 4 : 8) * t.fft_len) * sizeof(cf32); // 4 x N arrays: pilot values (+ parked data symbols)
    if (fused) {
        b += size_t((t.fft_len + 3) & ~3) * sizeof(int16_t);              // data-carrier positions in serializer order
        b += size_t(rho0 + 1) * size_t(t.max_occ) * sizeof(cf32);         // parked rows
    }
    return (b + 15) & ~size_t(15);
}

size_t payload_lds_bytes(const RxTables& t) { return frame_eq_lds_bytes(t, false, false); }

static bool payload_staged(const RxTables& t, bool eq_rows)
{
    // the staged form (no global memory inside an equaliser step) serves the usual configuration when its LDS fits
    return t.n_occ_sets == 1 && t.n_pilot_sets <= 32 && !eq_rows && frame_eq_lds_bytes(t, true, false) <= 64 * 1024;
}

// Fused phase B: how many of the frame's OFDM symbols are parked in LDS (rho0 + 1, rho0 a multiple of 4) -- as many as
// leave the CU's 160 KB of LDS to the four workgroups per SIMD wave slot the kernel's registers allow; the remaining
// L - rho0 source rows (at most kEqRegSlots) stay in registers.  -1: the frame does not fit (two-kernel form).
static int fused_rho0(const RxTables& t)
{
    const int L = t.frame_length;
    if (L > kEqFusedMaxSyms) return -1;
    const size_t waves = size_t((t.fft_len + 63) / 64);
    size_t budget = std::min<size_t>(64 * 1024, size_t(160 * 1024) * waves / 16);
    if (const char* e = getenv("DTL_FUSED_LDS_BUDGET")) budget = size_t(atol(e));   // experiments
    const size_t fixed = frame_eq_lds_bytes(t, true, true, -1) + 640 /* static: constellation, reciprocals */ + 1024 /* slack */;
    const size_t per_row = size_t(t.max_occ) * sizeof(cf32);
    if (budget < fixed + per_row) return -1;
    const int rows = int((budget - fixed) / per_row);
    int rho0 = ((rows - 1) / 4) * 4;                  // rho0 + 1 <= rows
    if (rho0 >= L) rho0 = (L + 3) & ~3;               // everything parked
    if (L - rho0 > kEqRegSlots) return -1;
    return rho0;
}

// Does phase B run as ONE kernel (frame equaliser with the soft demapper in its epilogue)?  Opt-in
// (DTL_RX_FUSED_PHASE_B=1) for FEC configurations whose frames fit on chip: measured on C3 the fused kernel moves 4.5 GB
// less per step through HBM but takes 3 % longer than the two kernels it replaces -- both are bound by the VALU issue
// rate, not by latency or memory, so there are no idle slots for the demapper to fill (DESIGN.md 10,
// profiles/r03/fused_eq_demap_*).  tests/test_fused_phase_b_gpu.py holds the two forms bit-identical.
bool payload_is_fused(const RxTables& t, bool eq_rows)
{
    const char* e = getenv("DTL_RX_FUSED_PHASE_B");
    if (!e || e[0] != '1') return false;
    return t.has_fec && payload_staged(t, eq_rows) && fused_rho0(t) >= 0;
}

// phase B: frame equaliser (FFT + decision-directed one-tap equaliser + SNR estimate) -> soft demapper (FEC) or
// hard demapper + packer + CRC (no FEC).  Fused form: one kernel, `soft_syms` written only when the caller wants the
// serializer's stream (soft_syms_wanted); two-kernel form: the equalised data symbols are handed over in `soft_syms`
// (n_frames x capacity), which must exist.
int launch_payload(const RxTables& t, const cf32* iq, int64_t n, const FrameDesc* frames, int n_frames,
                   const CfoSegment* segs, int n_segs, const cf32* taps, FrameStat* stats,
                   float* llr, cf32* eq_hard, cf32* eq_soft, cf32* soft_syms, bool soft_syms_wanted,
                   uint8_t* nofec_bytes, int32_t* nofec_info, int nofec_stride, hipStream_t st,
                   const FrameDesc* frames_host, hipEvent_t between)
{
    if (n_frames <= 0) return DTL_OK;
    // frames_host: the descriptors in pinned, coherent host memory -- the frame equaliser reads them from there (32 bytes
    // per workgroup) and leaves the copy in `frames` for the kernels behind it: no host-to-device copy in the stream (every
    // small copy behind a kernel costs ~12 us of device idle plus its own time)
    const FrameDesc* src = frames_host ? frames_host : frames;
    FrameDesc* copy = frames_host ? const_cast<FrameDesc*>(frames) : nullptr;
    const bool eq_rows = eq_hard || eq_soft;
    const bool staged = payload_staged(t, eq_rows);
    const bool fused = payload_is_fused(t, eq_rows);
    const int rho0 = fused ? fused_rho0(t) : 0;
    if (!fused && !soft_syms) { set_last_error("payload: no buffer for the equalised symbols"); return DTL_ERR_INVALID_ARG; }
    const size_t lds = frame_eq_lds_bytes(t, staged, fused, rho0);
    if (lds > 64 * 1024) { set_last_error("payload kernel: FFT scratch does not fit LDS"); return DTL_ERR_UNSUPPORTED; }
    const int mode = demap_mode(t.llr_dist_mode);
    if (fused) {
        cf32* ss = soft_syms_wanted ? soft_syms : nullptr;
        if (mode == 1)
            hipLaunchKernelGGL((frame_eq_kernel<true, false, 1>), dim3(n_frames), dim3(t.fft_len), lds, st, t, iq, n, src, segs, n_segs,
                               taps, stats, eq_hard, eq_soft, ss, llr, rho0, copy);
        else if (mode == 2)
            hipLaunchKernelGGL((frame_eq_kernel<true, false, 2>), dim3(n_frames), dim3(t.fft_len), lds, st, t, iq, n, src, segs, n_segs,
                               taps, stats, eq_hard, eq_soft, ss, llr, rho0, copy);
        else
            hipLaunchKernelGGL((frame_eq_kernel<true, false, 0>), dim3(n_frames), dim3(t.fft_len), lds, st, t, iq, n, src, segs, n_segs,
                               taps, stats, eq_hard, eq_soft, ss, llr, rho0, copy);
        DTL_HIP_CHECK(hipGetLastError());
        return DTL_OK;
    }
    if (staged)
        hipLaunchKernelGGL(frame_eq_kernel<true>, dim3(n_frames), dim3(t.fft_len), lds, st, t, iq, n, src, segs, n_segs, taps,
                           stats, eq_hard, eq_soft, soft_syms, nullptr, 0, copy);
    else
        hipLaunchKernelGGL(frame_eq_kernel<false>, dim3(n_frames), dim3(t.fft_len), lds, st, t, iq, n, src, segs, n_segs, taps,
                           stats, eq_hard, eq_soft, soft_syms, nullptr, 0, copy);
    if (between) DTL_HIP_CHECK(hipEventRecord(between, st));
    if (t.has_fec) {
        const dim3 grid(n_frames, (t.frame_capacity + 255) / 256);
        if (mode == 1) hipLaunchKernelGGL(demap_kernel<1>, grid, dim3(256), 0, st, t, frames, stats, soft_syms, llr);
        else if (mode == 2) hipLaunchKernelGGL(demap_kernel<2>, grid, dim3(256), 0, st, t, frames, stats, soft_syms, llr);
        else hipLaunchKernelGGL(demap_kernel<0>, grid, dim3(256), 0, st, t, frames, stats, soft_syms, llr);
    } else {
        hipLaunchKernelGGL(nofec_tail_kernel, dim3(n_frames), dim3(256), 0, st, t, frames, soft_syms, nofec_bytes, nofec_info,
                           nofec_stride);
    }
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

int launch_tb_check(const TbGather* tbs, int n_tbs, const uint8_t* bits, const int32_t* nit, uint8_t* staging,
                    int stride, TbResult* res, int32_t* len, hipStream_t st, bool packed)
{
    if (n_tbs <= 0) return DTL_OK;
    const size_t lds = (256 + 16 * 32) * 4 + size_t((stride + 15) & ~15);
    if (lds > 64 * 1024) { set_last_error("tb_check: transport block too large"); return DTL_ERR_UNSUPPORTED; }
    // zero-append maps of the CRC-32 register for 2^j bytes, j = 0..15 (built once per process and device)
    static uint32_t* d_zmat = nullptr;
    static int zmat_dev = -1;
    static std::mutex zmat_mu;
    std::lock_guard<std::mutex> zmat_lock(zmat_mu);
    int dev = 0;
    DTL_HIP_CHECK(hipGetDevice(&dev));
    if (!d_zmat || zmat_dev != dev) {
        uint32_t T[256];
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            T[i] = c;
        }
        std::vector<uint32_t> z(16 * 32);
        for (int b = 0; b < 32; b++) { const uint32_t v = 1u << b; z[b] = T[v & 0xFF] ^ (v >> 8); } // one zero byte
        for (int j = 1; j < 16; j++)
            for (int b = 0; b < 32; b++) { // the map for 2^j bytes is the square of the one for 2^(j-1)
                uint32_t v = z[(j - 1) * 32 + b], acc = 0;
                for (int c = 0; c < 32; c++) if ((v >> c) & 1u) acc ^= z[(j - 1) * 32 + c];
                z[j * 32 + b] = acc;
            }
        z.insert(z.end(), T, T + 256);                // the byte table behind the maps (tb_check_kernel reads both)
        uint32_t* p = nullptr;
        DTL_HIP_CHECK(hipMalloc(&p, z.size() * 4));
        DTL_HIP_CHECK(hipMemcpy(p, z.data(), z.size() * 4, hipMemcpyHostToDevice));
        d_zmat = p; zmat_dev = dev;
    }
    if ((stride / 1) >= (1 << 16)) { set_last_error("tb_check: transport block above 64 KiB"); return DTL_ERR_UNSUPPORTED; }
    if (packed) hipLaunchKernelGGL(tb_check_kernel<true>, dim3(n_tbs), dim3(64), lds, st, tbs, n_tbs, bits, nit, staging, stride, res, d_zmat);
    else hipLaunchKernelGGL(tb_check_kernel<false>, dim3(n_tbs), dim3(64), lds, st, tbs, n_tbs, bits, nit, staging, stride, res, d_zmat);
    hipLaunchKernelGGL(tb_lengths_kernel, dim3((n_tbs + 255) / 256), dim3(256), 0, st, res, n_tbs, len);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

// soft bits of irregular transport blocks -> contiguous scratch runs (one workgroup per piece)
__global__ __launch_bounds__(256) void llr_gather_kernel(const LlrCopy* __restrict__ ops, int n_ops, float* __restrict__ llr)
{
    if (int(blockIdx.x) >= n_ops) return;
    const LlrCopy op = ops[blockIdx.x];
    float* dst = llr + op.dst;
    if (op.src == 0xFFFFFFFFu) {
        for (int i = threadIdx.x; i < op.len; i += 256) dst[i] = 0.0f;
    } else {
        const float* src = llr + op.src;
        for (int i = threadIdx.x; i < op.len; i += 256) dst[i] = src[i];
    }
}

int launch_llr_gather(const LlrCopy* ops, int n_ops, float* llr, hipStream_t st)
{
    if (n_ops <= 0) return DTL_OK;
    hipLaunchKernelGGL(llr_gather_kernel, dim3(n_ops), dim3(256), 0, st, ops, n_ops, llr);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

int launch_expand_cw(const TbGather* tbs, int n_tbs, CwDesc* out, hipStream_t st)
{
    if (n_tbs <= 0) return DTL_OK;
    hipLaunchKernelGGL(expand_cw_kernel, dim3(n_tbs), dim3(64), 0, st, tbs, n_tbs, out);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

int launch_nofec_lengths(const int32_t* info, int n, int32_t* len, hipStream_t st)
{
    if (n <= 0) return DTL_OK;
    hipLaunchKernelGGL(nofec_lengths_kernel, dim3((n + 255) / 256), dim3(256), 0, st, info, n, len);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

int launch_compact(const uint8_t* staging, int stride, const int32_t* len, int n, uint32_t* offs,
                   uint32_t* total, uint8_t* out, hipStream_t st)
{
    if (n <= 0) { DTL_HIP_CHECK(hipMemsetAsync(total, 0, sizeof(uint32_t), st)); return DTL_OK; }
    hipLaunchKernelGGL(scan_lengths_kernel, dim3(1), dim3(1024), 0, st, len, n, offs, total);
    hipLaunchKernelGGL(compact_kernel, dim3(n), dim3(128), 0, st, staging, stride, len, offs, n, out);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}


// The channel estimator's and the equalisers' arithmetic (Smith's complex division, the offset search's ordering) has
// been compared with the oracle's on NaN, +-0 and every NORMAL float up to 2^62 in magnitude (tests/test_fuzz_gpu.py:
// tiny normals such as 1e-30 included); with denormal components, infinities or |x| >= 1e20 (squares overflow) both sides
// return garbage, but not the same garbage.  The per-block entry points (dtl_chanest, dtl_header_decode,
// dtl_equalize_frames) refuse such input instead.
__global__ __launch_bounds__(256) void range_check_kernel(const float* __restrict__ x, int64_t n, unsigned int* __restrict__ bad)
{
    unsigned int mine = 0;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        const float a = fabsf(x[i]);
        const bool fine = a != a || a == 0.0f || (a >= 0x1p-126f && a <= 0x1p62f);
        mine += fine ? 0u : 1u;
    }
    if (mine) atomicAdd(bad, 1u);   // (the caller only tests the sum against zero)
}

int launch_range_check(const float* x, int64_t n, unsigned int* d_bad, hipStream_t st)
{
    if (n <= 0) return DTL_OK;
    const int64_t blocks = std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(range_check_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, d_bad);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

// ---- launchers of the stand-alone stages
void constellation_points_host(int cnst, cf32* out16)
{
    for (int i = 0; i < 16; i++) out16[i] = { 0.0f, 0.0f };
    switch (cnst) {
    case 1: out16[0] = { -1, 0 }; out16[1] = { 1, 0 }; break;
    case 2: {
        const float s = float(double(float(0.707107)) * 0.5); // SQRT_TWO macro, lib/dtl/constellation.cc:19-23
        out16[0] = { -s, -s }; out16[1] = { s, -s }; out16[2] = { -s, s }; out16[3] = { s, s };
    } break;
    case 3: {
        const float angle = float(M_PI / 8.0);
        const int mult[8] = { 1, 7, 15, 9, 3, 5, 13, 11 };
        for (int i = 0; i < 8; i++) out16[i] = { float(std::cos(mult[i] * angle)), float(std::sin(mult[i] * angle)) };
    } break;
    case 4: {
        const float level = std::sqrt(float(0.1));
        for (int i = 0; i < 16; i++)
            out16[i] = { ((i & 2) ? 3 : 1) * level * ((i & 4) ? -1.f : 1.f), ((i & 1) ? 3 : 1) * level * ((i & 8) ? 1.f : -1.f) };
    } break;
    default: break;
    }
}

int launch_soft_demap(const cf32* syms, int64_t n, int cnst, double noise, int dist_mode, float* llr, hipStream_t st)
{
    if (n <= 0) return DTL_OK;
    if (cnst < 1 || cnst > 4) { set_last_error("Unknown constellation"); return DTL_ERR_INVALID_ARG; }
    CnstPoints pts;
    constellation_points_host(cnst, pts.p);
    // soft_cf_impl.cc:89-103: sigma = sqrt(noise), or 0.001 when no estimate is available
    double sigma = 0.001;
    if (!(noise < 0)) sigma = std::sqrt(noise); // negative: no estimate; a NaN estimate passes through like any tag value
    const float npwr = float(sigma), rinv = float(1.0 / double(npwr));
    const int64_t grid = (n + 255) / 256;
    const int mode = demap_mode(dist_mode);
    if (mode == 1) hipLaunchKernelGGL(soft_demap_kernel<1>, dim3((unsigned)grid), dim3(256), 0, st, syms, n, cnst, pts, npwr, rinv, llr);
    else if (mode == 2) hipLaunchKernelGGL(soft_demap_kernel<2>, dim3((unsigned)grid), dim3(256), 0, st, syms, n, cnst, pts, npwr, rinv, llr);
    else hipLaunchKernelGGL(soft_demap_kernel<0>, dim3((unsigned)grid), dim3(256), 0, st, syms, n, cnst, pts, npwr, rinv, llr);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

int launch_fft_batch(const cf32* in, cf32* out, int64_t n_vec, int N, int forward, hipStream_t st)
{
    if (n_vec <= 0) return DTL_OK;
    if (N < 4 || N > 2048 || (N & (N - 1))) { set_last_error("fft_batch: fft_len must be a power of two in [4, 2048]"); return DTL_ERR_INVALID_ARG; }
    // twiddle / bit-reversal tables per transform length, built once per process and device
    struct Tab { int N, dev; cf32* tw; uint16_t* br; };
    static std::vector<Tab> tabs;
    static std::mutex tabs_mu;
    std::lock_guard<std::mutex> tabs_lock(tabs_mu);
    int dev = 0;
    DTL_HIP_CHECK(hipGetDevice(&dev));
    const Tab* t = nullptr;
    for (const auto& e : tabs) if (e.N == N && e.dev == dev) t = &e;
    if (!t) {
        std::vector<cf32> tw(N / 2);
        for (int k = 0; k < N / 2; k++) { const double ang = -2.0 * M_PI * k / N; tw[k] = { float(std::cos(ang)), float(std::sin(ang)) }; }
        std::vector<uint16_t> br(N);
        int logn = 0;
        while ((1 << logn) < N) logn++;
        for (int i = 0; i < N; i++) { int r = 0; for (int b = 0; b < logn; b++) if (i & (1 << b)) r |= 1 << (logn - 1 - b); br[i] = uint16_t(r); }
        Tab e{ N, dev, nullptr, nullptr };
        DTL_HIP_CHECK(hipMalloc(&e.tw, tw.size() * sizeof(cf32)));
        DTL_HIP_CHECK(hipMalloc(&e.br, br.size() * 2));
        DTL_HIP_CHECK(hipMemcpy(e.tw, tw.data(), tw.size() * sizeof(cf32), hipMemcpyHostToDevice));
        DTL_HIP_CHECK(hipMemcpy(e.br, br.data(), br.size() * 2, hipMemcpyHostToDevice));
        tabs.push_back(e);
        t = &tabs.back();
    }
    const size_t lds = size_t(N + N / 2) * sizeof(cf32);
    hipLaunchKernelGGL(fft_batch_kernel, dim3((unsigned)n_vec), dim3(N / 2), lds, st, in, out, N, forward, t->tw, t->br);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

static int crc_zero_maps(uint32_t** out, int levels)
{
    static uint32_t* d_z = nullptr;
    static int z_dev = -1;
    static std::mutex z_mu;
    std::lock_guard<std::mutex> z_lock(z_mu);
    int dev = 0;
    DTL_HIP_CHECK(hipGetDevice(&dev));
    if (!d_z || z_dev != dev) {
        uint32_t T[256];
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            T[i] = c;
        }
        std::vector<uint32_t> z(32 * 32);
        for (int b = 0; b < 32; b++) { const uint32_t v = 1u << b; z[b] = T[v & 0xFF] ^ (v >> 8); }
        for (int j = 1; j < 32; j++)
            for (int b = 0; b < 32; b++) {
                uint32_t v = z[(j - 1) * 32 + b], acc = 0;
                for (int c = 0; c < 32; c++) if ((v >> c) & 1u) acc ^= z[(j - 1) * 32 + c];
                z[j * 32 + b] = acc;
            }
        uint32_t* p = nullptr;
        DTL_HIP_CHECK(hipMalloc(&p, z.size() * 4));
        DTL_HIP_CHECK(hipMemcpy(p, z.data(), z.size() * 4, hipMemcpyHostToDevice));
        d_z = p; z_dev = dev;
    }
    (void)levels;
    *out = d_z;
    return DTL_OK;
}

int launch_crc32(const uint8_t* bytes, int64_t len, uint32_t* d_out, hipStream_t st)
{
    if (len < 0 || len > int64_t(0xFFFFFFFFu)) { set_last_error("crc32: at most 2^32-1 bytes"); return DTL_ERR_INVALID_ARG; }
    uint32_t* z = nullptr;
    int rc = crc_zero_maps(&z, 32);
    if (rc != DTL_OK) return rc;
    hipLaunchKernelGGL(crc32_kernel, dim3(1), dim3(64), 0, st, bytes, len, z, d_out);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

} // namespace dtl
