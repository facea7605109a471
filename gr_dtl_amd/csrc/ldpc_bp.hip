// ldpc_bp.hip -- K10: LDPC sum-product belief propagation, one 64-lane wavefront per codeword.
//
// Replaces gr::dtl::ldpc_dec::decode (lib/dtl/ldpc_dec.cc:62-71) and the GNU Radio awgn_bp
// decoder it drives.  Arithmetic is the reference's: likelihood-ratio domain, IEEE double,
// same operand order per check row (mlist order) and per variable column (nlist order), so the
// hard decisions and iteration counts equal the CPU decoder's.  No MFMA: there is no dense
// contraction here, only 4 double divisions per Tanner-graph edge per iteration.
//
// Mapping to CDNA4:
//  * one wavefront == one codeword; all messages of the codeword live in LDS (E doubles,
//    7.1 KB for the (300,152) code) and are updated in place;
//  * work is EDGE-parallel (lane == edge) so the 64 lanes stay ~93-99 % busy for any row/column
//    weight; whole rows (check phase) / whole variables (variable phase) are packed into one
//    64-lane pass so that an in-place update can never race with a sibling lane's read;
//  * the syndrome is a per-check XOR of LDS bytes reduced with a 64-bit ballot;
//  * a persistent grid strides over the codeword batch (codewords have 0..max_it iterations,
//    the stride averages the imbalance); 16 single-wave workgroups fit a CU's 160 KB LDS.
#include "ldpc_bp.h"
#include "dtl_math.h"

#include <type_traits>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#pragma clang fp contract(off)

#include "dtl_ablate.h"   // timing experiments: every `ablated(k)` below is false in the shipped library

namespace dtl {

namespace {

constexpr int kWave = 64;
constexpr float kShortenedValue = -15.0f; // lib/dtl/tb_decoder.cc:145

__device__ __forceinline__ bool syndrome_is_zero(const LdpcTables& t, const uint8_t* est, int lane)
{
    unsigned bad = 0;
    for (int c = lane; c < t.m; c += kWave) {
        unsigned s = 0;
        const int e1 = t.row_ptr[c + 1];
        for (int e = t.row_ptr[c]; e < e1; e++) s ^= est[t.edge_var[e]];
        bad |= s;
    }
    return __ballot(bad != 0) == 0ull;
}

__global__ __launch_bounds__(kWave) void ldpc_bp_kernel(LdpcTables t, const float* __restrict__ llr,
                                                       const CwDesc* __restrict__ desc, int ncw,
                                                       int max_it, float sigma,
                                                       uint8_t* __restrict__ out_bits,
                                                       int32_t* __restrict__ out_nit)
{
    extern __shared__ double smem[];
    double* msg = smem;                                  // n_edges doubles: Q / t / R in place
    double* rx = msg + t.n_edges;                        // n doubles: channel likelihood ratios
    uint8_t* est = reinterpret_cast<uint8_t*>(rx + t.n); // n bytes: hard decisions
    float* ybuf = reinterpret_cast<float*>(msg);         // n floats, only before msg is live

    const int lane = threadIdx.x;
    const int ncheck = t.n - t.k;

    for (int cw = blockIdx.x; cw < ncw; cw += gridDim.x) {
        uint32_t off0, off1, out_off;
        int len0, len1, k_used;
        if (desc) {
            CwDesc d = desc[cw];
            off0 = d.llr_off0; off1 = d.llr_off1; out_off = d.out_off; len0 = d.len0; len1 = d.len1; k_used = d.k_used;
        } else {
            off0 = uint32_t(cw) * uint32_t(t.n); off1 = 0; out_off = uint32_t(cw) * uint32_t(t.k);
            len0 = t.n; len1 = 0; k_used = t.k;
        }
        const int n_sent = ncheck + k_used;

        // y[permute[i]] = -llr[i]; shortened positions were never sent: -15 (tb_decoder.cc:163-183)
        for (int v = lane; v < t.n; v += kWave) {
            const int src = t.inv_perm[v];
            float val = kShortenedValue;
            if (src < n_sent) {
                if (src < len0) val = llr[size_t(off0) + src];
                else val = (src < len0 + len1) ? llr[size_t(off1) + (src - len0)] : 0.0f; // else: never received
            }
            const float y = -1 * val;
            ybuf[v] = y;
            est[v] = (y < 0) ? 1 : 0; // awgn_bp::compute_init_estimate
        }
        __syncthreads();

        int nit = 0;
        if (!syndrome_is_zero(t, est, lane)) {
            // awgn_bp::rx_lr_calc
            // (float expression, exp of the float-rounded argument; rx does not overlap ybuf)
            const float s2 = sigma * sigma;
            for (int v = lane; v < t.n; v += kWave) rx[v] = dm_exp64(double((-1 * 2 * ybuf[v]) / s2));
            __syncthreads(); // ybuf (aliasing msg) is dead from here
            // awgn_bp::spa_initialize
            for (int e = lane; e < t.n_edges; e += kWave) msg[e] = rx[t.edge_var[e]];
            __syncthreads();

            while (nit < max_it) {
                nit += 1;
                // ---- awgn_bp::update_chks, part a: t_e = 2/(1+Q_e) - 1 for every edge
#pragma unroll 2
                for (int e = lane; e < t.n_edges; e += kWave) {
                    const double q = msg[e];
                    msg[e] = 2 / (1 + q) - 1;
                }
                __syncthreads();
                // ---- part b: row product in mlist order, extrinsic ratio, R_e
                for (int p = 0; p < t.n_chk_pass; p++) {
                    const ChkSlot s = t.chk[p * kWave + lane];
                    double r_new = 0.0;
                    if (s.edge != 0xFFFF) {
                        double product = 1.0;
                        for (int j = 0; j < s.deg; j++) product = product * msg[s.row_start + j];
                        const double prd = product / msg[s.edge];
                        r_new = (1 - prd) / (1 + prd);
                    }
                    __syncthreads(); // every lane of the row has read before anyone writes
                    if (s.edge != 0xFFFF) msg[s.edge] = r_new;
                }
                __syncthreads();
                // ---- awgn_bp::update_vars + decision
                for (int p = 0; p < t.n_var_pass; p++) {
                    const VarSlot s = t.var[p * kWave + lane];
                    double q_new = 0.0;
                    if (s.edge != 0xFFFF) {
                        double sum = rx[s.var];
                        for (int i = 0; i < s.deg; i++) sum = sum * msg[t.col_edge[s.col_start + i]];
                        q_new = sum / msg[s.edge];
                        if (s.first) est[s.var] = (sum > 1) ? 1 : 0;
                    }
                    __syncthreads();
                    if (s.edge != 0xFFFF) msg[s.edge] = q_new;
                }
                __syncthreads();
                if (syndrome_is_zero(t, est, lane)) break;
            }
        }

        // cldpc::get_systematic_bits; only the k_used bits that were sent are kept
        for (int i = lane; i < k_used; i += kWave) out_bits[size_t(out_off) + i] = est[t.sys_src[i]];
        if (lane == 0) out_nit[cw] = nit;
        __syncthreads(); // LDS is reused by the next codeword
    }
}


// ------------------------------------------------------------------------------------------
// Node-parallel variant (the one normally launched): lane == check node in the check phase,
// lane == variable node in the variable phase.  Each lane then owns 6 (resp. 3) INDEPENDENT
// division chains -- the latency of one f64 division hides behind the others -- and no slot
// tables, no redundant row reads and no intermediate t_e store are needed: 2x fewer wave
// instructions than the edge-parallel form for the same IEEE operations in the same order.
// A 1024-thread workgroup keeps 16 codewords (one per wave) in one CU's 160 KB LDS next to a
// single copy of the code's index tables; waves never synchronise with each other.
// Variables are numbered in ON-AIR order (variable i here == the reference's variable permute[i]): soft bits are
// used where they land (lane l of load p holds variable l + 64 p), hard decisions are NV 64-bit ballot words, the
// syndrome is parity(row bitmap AND decisions), and the data bits are that bitmap shifted down by `rank` bits -- no
// scatter through the permutation, no per-variable decision bytes, no lookups of the systematic positions.
// The kernel is bound by instruction issue (measured on gfx950: ~one wave instruction of ANY kind per SIMD per 4
// cycles -- instructions/codeword x 4 cycles x codewords/SIMD reproduces the kernel time both for clean codewords
// and for the receive chain's mix, profiles/r03/ldpc_zero_iteration.txt), so every instruction removed counts the
// same, scalar or vector, and memory waits matter only in that they must not serialise.
// ------------------------------------------------------------------------------------------
constexpr int kNpWaves = 16;
constexpr int kNpChunk = 16;  // codewords per queue ticket (large batches)
constexpr int kNpQueues = 8;       // ticket counters per launch ...
constexpr int kNpQueueStride = 16; // ... this many 32-bit words apart (one 64-byte line each)
constexpr int kNpSplitIterations = 3; // decisions-first variable passes (exact arithmetic): measured, tools/ldpc_probe.py

__device__ __forceinline__ void wave_sync()
{
    // one wave owns its LDS slice: DS operations of a wave execute in order, only the compiler
    // has to be kept from moving memory operations across the phase boundary
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Index tables of the node-parallel kernel use FIXED strides: check c owns message slots j*m + c, j < MAXR -- entry-major,
// so that the lanes of a check pass (lane == check) touch consecutive doubles: conflict-free LDS reads and writes; a
// check-major row of 6 doubles (48 bytes) puts lanes l and l+16 on one bank --
// (rows shorter than MAXR are padded), variable v owns four 16-bit slot indices (columns lighter than four
// entries are padded).  Padding costs no branch: a padded row slot names the dummy variable n, whose channel
// LR is 0 and whose hard decision is 0 -- its message stays 0, i.e. the factor 2/(1+0)-1 == 1 in the row
// product -- and a padded column entry names the message slot kept at 1.0 (factor 1 in the column product)
// (set back to 1.0 behind the writes of every variable pass).  Multiplying by exactly 1.0 changes no bit of a product, so
// the arithmetic is still the reference's, without per-edge degree tests and without row/column pointer reads.
// FAST (dtl_ldpc_set_arith(h, DTL_ARITH_FAST); NOT the reference's bits -- oracle: orc_ldpc_decode_fast): the same
// recursion with ONE division per batch of a row's quotients (prefix products, one reciprocal, walk back) and prefix x
// suffix extrinsic products instead of (full product) / own factor: 2 divisions per check per iteration instead of
// 4 per edge.  The channel LLR (+-40) and the check-node products (+-(1 - 2^-20)) are clamped: check-to-variable
// ratios then lie in [2^-21, 2^21], variable-to-check ratios below 2^121 for column weights <= 4, a row's product
// of (1 + Q) below 2^968 for row weights <= 8 -- every intermediate stays finite (the reference's form saturates to
// 0 / inf and then NaN, whose decision is 0 whatever was sent).
constexpr float kFastLlrMax = 40.0f;
constexpr double kFastPMax = 1.0 - 0x1p-20;

// out[j] = num[j] / den[j], j < D, with one division: P_j = den_0 .. den_j, I = 1 / P_{D-1}, then
// 1 / den_j = I P_{j-1}, I <- I den_j walking back (the operation order of the oracle's batch_quotients)
template <int D>
__device__ __forceinline__ void batch_quotients(const double (&num)[D], const double (&den)[D], double (&out)[D])
{
    double P[D];
    P[0] = den[0];
#pragma unroll
    for (int j = 1; j < D; j++) P[j] = P[j - 1] * den[j];
    double I = 1 / P[D - 1];
#pragma unroll
    for (int j = D - 1; j >= 1; j--) {
        out[j] = num[j] * (I * P[j - 1]);
        I = I * den[j];
    }
    out[0] = num[0] * I;
}

// a / b for the quotient (row product) / (own factor) of the exact check update: the IEEE double division sequence
// the compiler emits (v_rcp_f64, two Newton steps, quotient, residual, final fused multiply-add) WITHOUT
// v_div_scale / v_div_fixup -- same bits whenever those two would do nothing.  Here they never do: a factor is
// t = 2/(1+q) - 1, so 0 or 2^-53 <= |t| <= 2^54 (or +-inf / NaN), the product of at most 8 of them lies in
// [2^-424, 2^432]: no operand or quotient is near the over/underflow ranges the scaling handles, and the only
// special cases are 0/0, inf/inf and NaN operands, which come out as NaN from this sequence as from the full one,
// a zero product over a non-zero factor, which comes out as +0 where IEEE gives +-0, and an infinite product (a
// factor 2/0 - 1) over a finite factor, which comes out as NaN where IEEE gives inf -- neither is observable: the
// quotient's one consumer is R = (1 - prd) / (1 + prd), which is 1 resp. NaN either way.
// (tools/bp_div_check.hip compares through R: 5 x 10^10 operand pairs per division site, profiles/r02/bp_div_check.txt.)
__device__ __forceinline__ double div_row_product(double a, double b)
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

// a / b, IEEE double.  TAME == false: the compiler's sequence (v_div_scale x 2, v_rcp_f64, Newton, v_div_fmas,
// v_div_fixup: 11 instructions).  TAME == true: the same sequence without the two v_div_scale and with a plain
// fused multiply-add for v_div_fmas (9 instructions) -- v_div_fixup still settles zero / infinite / NaN operands.
// Scaling only moves operands away from the over/underflow ranges by exact powers of two, so the two forms give the
// same bits whenever no intermediate of the unscaled one leaves the normal range.  For a codeword whose channel
// exponents all satisfy |arg| <= 500 ("tame": rx in [2^-722, 2^722]) that holds for every division of the exact
// update: every finite check-to-variable ratio is 0 or has magnitude in [2^-54, 2^54] (a quotient of (1 -+ prd) with
// prd a multiple of 2^-53 near +-1, or near -1 for |prd| large), so a column product is 0 or within
// [2^-884, 2^884], a variable-to-check ratio within [2^-938, 2^938], 1 + Q is 0 or within [2^-53, 2^939], and the
// reciprocals, quotients and residuals (>= 2^-53 x the numerator) of all of them are normal numbers.
template <bool TAME>
__device__ __forceinline__ double div_exact(double a, double b)
{
    if constexpr (!TAME) {
        return a / b;
    } else {
        double r = __builtin_amdgcn_rcp(b);
        double e = __builtin_fma(-b, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-b, r, 1.0);
        r = __builtin_fma(r, e, r);
        const double q = a * r;
        const double rem = __builtin_fma(-b, q, a);
        return __builtin_amdgcn_div_fixup(__builtin_fma(rem, r, q), b, a);
    }
}
constexpr float kTameArgMax = 500.0f;

template <int MAXR, int MAXC, int NV, bool ROWS_FULL, bool FAST>
__global__ __launch_bounds__(kNpWaves * kWave) void ldpc_bp_np_kernel(LdpcTables t, NpShared sh,
                                                                     const uint16_t* __restrict__ chk_var_g,
                                                                     const uint16_t* __restrict__ var_edge_g,
                                                                     const uint8_t* __restrict__ chk_deg_g,
                                                                     const unsigned long long* __restrict__ chk_bits_g,
                                                                     const float* __restrict__ llr,
                                                                     const CwDesc* __restrict__ desc, int ncw,
                                                                     int max_it, float sigma,
                                                                     uint8_t* __restrict__ out_bits,
                                                                     int32_t* __restrict__ out_nit,
                                                                     unsigned int* __restrict__ queue, int chunk, int packed, int split_iterations, int steal)
{
    static_assert(MAXC <= 4, "a variable's slot indices are read as one 64-bit word");
    extern __shared__ unsigned char np_smem[];
    uint16_t* chk_var = reinterpret_cast<uint16_t*>(np_smem + sh.chk_var);    // m * MAXR
    uint16_t* var_edge = reinterpret_cast<uint16_t*>(np_smem + sh.var_edge);  // n * 4
    uint8_t* chk_deg = np_smem + sh.chk_deg;                                  // m
    unsigned long long* chk_bits = reinterpret_cast<unsigned long long*>(np_smem + sh.chk_bits); // m x (NV | 1) row bitmaps
    const int n_slots = t.m * MAXR;
    for (int i = threadIdx.x; i < n_slots; i += blockDim.x) chk_var[i] = chk_var_g[i];
    for (int i = threadIdx.x; i < t.n * 4; i += blockDim.x) var_edge[i] = var_edge_g[i];
    for (int i = threadIdx.x; i < t.m; i += blockDim.x) chk_deg[i] = chk_deg_g[i];
    for (int i = threadIdx.x; i < t.m * (NV | 1); i += blockDim.x) chk_bits[i] = chk_bits_g[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* mine = np_smem + sh.wave_base + size_t(wave) * sh.wave_stride;
    double* msg = reinterpret_cast<double*>(mine);   // n_slots messages + [n_slots] == 1.0 (+ one spare)
    double* rx = msg + n_slots + 2;                  // n channel LRs + rx[n] == 0 (dummy variable)
    // hard decisions on their way out, 1 bit each: NV + 1 64-bit words (last: 0) over the first channel LRs, which are
    // dead once a codeword's iterations have ended and are written again only behind the next pass's `emit`
    uint32_t* xw = reinterpret_cast<uint32_t*>(rx);
    const int kOneSlot = n_slots;
    const int ncheck = t.n - t.k;
    const float s2 = sigma * sigma;

    // Hard decisions are NV uniform 64-bit words (X[p] bit l = decision of variable l + 64 p: one v_cmp each); row c of H
    // is NV 64-bit masks in LDS (check-major with an odd stride in words: one address per lane, constant offsets, and the
    // 64-bit reads of a pass fall on distinct banks), and the syndrome bit of a check is the parity of the AND of the
    // two -- no per-variable lookups.
    auto syndrome_ok = [&](const unsigned long long (&X)[NV]) -> bool {
        if constexpr (ablated(21)) return __ballot(X[0] == 0xdeadbeefull) == 0ull;
        // (a pass of 64 checks at a time, left at the first pass that holds an unsatisfied check: while a codeword has not
        // converged that is nearly always the first one)
        for (int c0 = 0; c0 < t.m; c0 += kWave) {
            const int c = min(c0 + lane, t.m - 1);        // (a lane past the last check repeats it)
            unsigned long long fold = 0;
#pragma unroll
            for (int p = 0; p < NV; p++) fold ^= chk_bits[c * (NV | 1) + p] & X[p];
            if (__ballot((__popc(unsigned(fold) ^ unsigned(fold >> 32)) & 1) != 0) != 0ull) return false;
        }
        return true;
    };
    // The memory pipeline of a wave: while codeword i is decoded, the soft bits of i + 1 are in flight and the
    // descriptor of i + 2 is being read -- through the SCALAR cache, as five whole dwords, so that waiting for it never
    // drains the vector-memory queue (a 16-bit field read makes the compiler use a vector load plus
    // s_waitcnt vmcnt(0), which waits for the soft bits just requested and for the previous codeword's stores: one
    // full memory latency per codeword, 3/4 of the time of a codeword that needs no iteration).
    struct Meta { uint32_t off0, off1, out_off; int len0, len1, k_used, n_sent, n_recv; };
    auto load_meta = [&](int cw, Meta& m) {
        if (desc) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(desc + cw);
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
            m.off0 = w0; m.off1 = w1; m.out_off = w2; m.len0 = int(w3 & 0xFFFFu); m.len1 = int(w3 >> 16); m.k_used = int(w4 & 0xFFFFu);
        } else {
            m.off0 = uint32_t(cw) * uint32_t(t.n); m.off1 = 0; m.out_off = uint32_t(cw) * uint32_t(t.k);
            m.len0 = t.n; m.len1 = 0; m.k_used = t.k;
        }
        m.n_sent = min(ncheck + m.k_used, t.n);        // beyond: never sent (shortened), tb_decoder.cc:163-183
        m.n_recv = min(m.n_sent, m.len0 + m.len1);     // beyond: sent but never received
    };
    // Soft bits of one codeword in on-air order, lane l holds positions l, l+64, ... (coalesced).  The loads are
    // unconditional -- a lane outside the received part reads the codeword's last received soft bit -- and their
    // values are put right (`settle`) only where the codeword becomes the current one: no branch and no use of a loaded
    // value near the loads, so the requests go out back to back and nothing waits for them here.  A codeword that
    // received NOTHING (len0 == len1 == 0: it lies in a gap) reads soft bit 0 of the stream, whatever its offsets say:
    // always a valid address of the caller's buffer (a descriptor with len0 == 0 and a stale llr_off1 used to be
    // dereferenced at off1).
    auto fetch = [&](const Meta& m, float (&raw)[NV]) {
        const unsigned last = unsigned(max(m.n_recv, 1) - 1);
        // position i of the codeword is soft bit off0 + i of the stream, or off1 + (i - len0) behind a frame boundary:
        // one 32-bit sum either way (offsets are 32-bit stream positions; the wrap-around of `hop` cancels)
        const unsigned len0 = unsigned(m.len0), hop = m.off1 - m.off0 - len0;
        const bool nothing = m.n_recv == 0;           // (wave-uniform)
#pragma unroll
        for (int p = 0; p < NV; p++) {
            const unsigned i = min(unsigned(lane + p * kWave), last);
            const uint32_t at = nothing ? 0u : m.off0 + i + (i < len0 ? 0u : hop);
            raw[p] = ablated(23) ? __uint_as_float(at | 0x80000000u) : llr[size_t(at)];
        }
    };
    auto settle = [&](const Meta& m, const float (&raw)[NV], float (&vals)[NV]) {
#pragma unroll
        for (int p = 0; p < NV; p++) {
            const int i = lane + p * kWave;
            vals[p] = i < m.n_recv ? raw[p] : (i < m.n_sent ? 0.0f : kShortenedValue);
        }
    };
    // Codewords are claimed `chunk` at a time from a global ticket counter: the iteration count
    // varies from 0 to max_it per codeword, so a static split would leave the slowest wave with up
    // to 30 % more work than the average.  (One ticket per codeword would serialise millions of
    // same-address device-scope atomics: measured 2.3x slower.)  The next codeword is known before
    // the current one is decoded so that its soft bits can be prefetched.
    // Tickets: `n_big` chunks of `chunk` codewords, then the last codewords in chunks of `kTailChunk` -- a wave that
    // takes a 16-codeword ticket near the end would still be decoding (~1 ms) when every other wave has run out of work
    constexpr int kTailChunk = 2;
    const int tail_cw = chunk > kTailChunk ? min(ncw, int(gridDim.x) * kNpWaves * 8) : 0;
    const int n_big = (ncw - tail_cw) / chunk;                  // codewords [0, n_big * chunk) in big chunks
    const int tail_first = n_big * chunk;
    const int n_tail = (ncw - tail_first + kTailChunk - 1) / kTailChunk;
    // Same-address device-scope atomics complete one every ~11.5 ns on this part (measured: a launch of 2^20
    // clean codewords took tickets x 11.5 ns whatever else the kernel did), so (a) the chunks are dealt round-robin to
    // kNpQueues counters in different cache lines -- a wave draws from its workgroup's home counter and moves on to the
    // next counter when one has run out (each counter is probed at most once past its end by a wave) -- and (b) a
    // wave asks for its NEXT ticket when it starts a chunk, not when it has finished it: the reply is back long before
    // it is needed (it returns in order with the soft-bit loads the wave waits for anyway).
    const int n_chunks = n_big + n_tail;
    int chunk_next = 0, chunk_end = 0;
    // steal == 0 (small batches): a wave draws from its workgroup's home counter only.  Every counter has its own waves (the
    // workgroups with blockIdx % kNpQueues == q), which drain it completely, so nothing is left behind; what is given up is
    // the balancing at the very end -- and what is saved are kNpQueues - 1 more same-address atomics per wave just to learn
    // that the other counters have run out too (4096 waves x 7 probes, serialised per counter: ~50 us of a launch that
    // otherwise takes 130 us for 4096 clean codewords)
    int home = int(blockIdx.x) % kNpQueues, queues_left = steal ? kNpQueues : 1;
    unsigned pending = 0;                                 // lane 0: the reply to the outstanding request
    auto request = [&]() {
        unsigned v = 0;
        if (lane == 0) v = atomicAdd(queue + home * kNpQueueStride, 1u);
        pending = v;
    };
    request();
    auto advance = [&]() -> int {
        if (chunk_next == chunk_end) {
            if (queues_left == 0) return ncw;
            for (;;) {
                const int g = int(__builtin_amdgcn_readfirstlane(pending)) * kNpQueues + home;   // chunk number
                if (g < n_chunks) {
                    if (g < n_big) {
                        chunk_next = g * chunk;
                        chunk_end = chunk_next + chunk;
                    } else {
                        chunk_next = tail_first + (g - n_big) * kTailChunk;
                        chunk_end = min(chunk_next + kTailChunk, ncw);
                    }
                    request();
                    break;
                }
                if (--queues_left == 0) return ncw;
                home = (home + 1) % kNpQueues;
                request();
            }
        }
        return chunk_next++;
    };
    auto check_phase = [&](auto first, auto tame) {
        constexpr bool FIRST = decltype(first)::value;
        constexpr bool TAME = decltype(tame)::value;
        // FIRST: the slots hold the factors t = 2/(1+Q) - 1 themselves (see the channel pass: Q_e of the first iteration is
        // the channel ratio of the edge's VARIABLE, so the factor is formed once per variable, not once per edge)
        // ---- update_chks: lane == check node.  When the last pass would fill at most half of the wave
        // (148 checks = 64 + 64 + 20), it runs with TWO lanes per check instead, each owning half of the
        // row: half the divisions per lane, and the row product is still formed in the reference's
        // order (the even lane multiplies its factors, hands the partial product to the odd lane, which
        // multiplies its own on top and hands the result back).
        int c0 = 0;
        for (; c0 + kWave <= t.m || t.m - c0 > kWave / 2; c0 += kWave) {
            const int c = c0 + lane;
            if (c < t.m) {
                double* row = msg + c;                    // message j of check c: row[j * t.m] (see the layout note)
                double tt[MAXR];
#pragma unroll
                for (int j = 0; j < MAXR; j++) tt[j] = FIRST ? row[j * t.m] : div_exact<TAME>(2.0, 1 + row[j * t.m]) - 1;
                double product = 1.0;
#pragma unroll
                for (int j = 0; j < MAXR; j++) product = product * tt[j];
                const int deg = ROWS_FULL ? MAXR : int(chk_deg[c]);
#pragma unroll
                for (int j = 0; j < MAXR; j++) {
                    const double prd = div_row_product(product, tt[j]);
                    const double val = div_exact<TAME>(1 - prd, 1 + prd);
                    row[j * t.m] = (ROWS_FULL || j < deg) ? val : 0.0; // a padded slot keeps the neutral message
                }
            }
        }
        if (c0 < t.m) {
            constexpr int HR = MAXR / 2;
            const int c = c0 + (lane >> 1);
            const bool upper = lane & 1;
            const bool valid = c < t.m;
            double* row = msg + (valid ? c : 0) + (upper ? HR : 0) * t.m;
            double tt[HR];
#pragma unroll
            for (int j = 0; j < HR; j++) tt[j] = FIRST ? row[j * t.m] : div_exact<TAME>(2.0, 1 + row[j * t.m]) - 1;
            double part = 1.0;
#pragma unroll
            for (int j = 0; j < HR; j++) part = part * tt[j];             // even lane: ((1 t0) t1) t2
            double product = __shfl_xor(part, 1);                         // odd lane: the even lane's partial
#pragma unroll
            for (int j = 0; j < HR; j++) product = product * tt[j];       // odd lane: ... t3) t4) t5
            const double back = __shfl_xor(product, 1);                   // even lane: the finished product
            if (!upper) product = back;
            const int deg = ROWS_FULL ? MAXR : int(chk_deg[valid ? c : 0]);
            if (valid) {
#pragma unroll
                for (int j = 0; j < HR; j++) {
                    const double prd = div_row_product(product, tt[j]);
                    const double val = div_exact<TAME>(1 - prd, 1 + prd);
                    row[j * t.m] = (ROWS_FULL || (upper ? HR : 0) + j < deg) ? val : 0.0;
                }
            }
        }
    };
    // FAST: lane == check node in every pass (the two-lanes-per-check tail of the exact form shares a division
    // sequence between lanes; here a row has only two divisions to begin with)
    auto check_phase_fast = [&](auto first) {
        constexpr bool FIRST = decltype(first)::value;
        for (int c0 = 0; c0 < t.m; c0 += kWave) {
            const int c = c0 + lane;
            if (c < t.m) {
                double* row = msg + c;
                const int deg = ROWS_FULL ? MAXR : int(chk_deg[c]);
                double a[MAXR], b[MAXR], tt[MAXR];
#pragma unroll
                for (int j = 0; j < MAXR; j++) {
                    const double q = FIRST ? rx[chk_var[c * MAXR + j]] : row[j * t.m];
                    a[j] = 1 + q;
                    b[j] = 1 - q;
                }
                batch_quotients<MAXR>(b, a, tt);          // t_j = (1 - Q_j) / (1 + Q_j)
                if (!ROWS_FULL) {
#pragma unroll
                    for (int j = 0; j < MAXR; j++) tt[j] = j < deg ? tt[j] : 1.0; // padding: exactly neutral
                }
                double pre[MAXR], suf[MAXR];
                pre[0] = 1.0;
#pragma unroll
                for (int j = 1; j < MAXR; j++) pre[j] = pre[j - 1] * tt[j - 1];
                suf[MAXR - 1] = 1.0;
#pragma unroll
                for (int j = MAXR - 2; j >= 0; j--) suf[j] = suf[j + 1] * tt[j + 1];
#pragma unroll
                for (int j = 0; j < MAXR; j++) {
                    double p = pre[j] * suf[j];
                    p = fmin(fmax(p, -kFastPMax), kFastPMax);
                    a[j] = (ROWS_FULL || j < deg) ? 1 + p : 1.0;
                    b[j] = 1 - p;
                }
                batch_quotients<MAXR>(b, a, tt);          // R_j = (1 - p_j) / (1 + p_j)
#pragma unroll
                for (int j = 0; j < MAXR; j++) row[j * t.m] = (ROWS_FULL || j < deg) ? tt[j] : 0.0;
            }
        }
    };
    // The stores of a codeword go out at the TOP of the next pass, just before the next soft bits are requested:
    // loads and stores share one in-order counter (s_waitcnt vmcnt), so the wait for the next codeword's soft bits at
    // the bottom of a pass also waits for every store issued before it -- stores issued at the bottom would be waited
    // for at once (a full write latency per codeword), stores issued at the top have had a whole pass to complete.
    // The decisions stay in the wave's LDS bitmap in between.
    int prev_cw = -1, prev_nit = 0, prev_k_used = 0;
    uint32_t prev_out_off = 0;
    auto emit = [&]() {
        const uint32_t out_off = prev_out_off;
        const int k_used = prev_k_used;
        if (packed) {
            // 1 bit per bit, LSB first (what fec_pack / the CRC check want) at bit position out_off of the zeroed output
            const bool bytewise = ((out_off | unsigned(k_used)) & 7u) == 0;  // the usual case: whole bytes, plain stores
            if (bytewise) {
                if (8 * lane < k_used) {                                     // k_used <= 512: one byte per lane
                    const int q = t.rank + 8 * lane;
                    const unsigned long long two = (unsigned long long)(xw[q >> 5]) | ((unsigned long long)(xw[(q >> 5) + 1]) << 32);
                    out_bits[size_t(out_off >> 3) + lane] = uint8_t(two >> (q & 31));
                }
            } else if (64 * lane < k_used) {
                // codewords of a block are not byte aligned in general: 64 bits per lane, OR-ed into the output words
                unsigned long long* words = reinterpret_cast<unsigned long long*>(out_bits);
                const int q = t.rank + 64 * lane, d = q >> 5, s = q & 31;
                const unsigned long long a = (unsigned long long)(xw[d]) | ((unsigned long long)(xw[d + 1]) << 32);
                const unsigned long long b = (unsigned long long)(xw[d + 1]) | ((unsigned long long)(xw[d + 2]) << 32);
                unsigned long long w = (unsigned long long)(uint32_t(a >> s)) | ((unsigned long long)(uint32_t(b >> s)) << 32);
                const int left = k_used - 64 * lane;
                if (left < 64) w &= (1ull << left) - 1ull;
                if (w) {
                    const unsigned long long pos = (unsigned long long)(out_off) + (unsigned long long)(64 * lane);
                    const unsigned sh = unsigned(pos & 63ull);
                    atomicOr(&words[pos >> 6], w << sh);
                    if (sh && (w >> (64 - sh))) atomicOr(&words[(pos >> 6) + 1], w >> (64 - sh));
                }
            }
        } else {
            for (int i = lane; i < k_used; i += kWave) {
                const int q = t.rank + i;
                out_bits[size_t(out_off) + i] = uint8_t((xw[q >> 5] >> (q & 31)) & 1u);
            }
        }
        if (lane == 0) out_nit[prev_cw] = prev_nit;
        wave_sync(); // the bitmap is rewritten by the next codeword
    };
    float cur[NV], nxt[NV];
    Meta m0{}, m1{}, m2{};
    int cw = advance();
    int cw1 = cw < ncw ? advance() : ncw;
    if (cw < ncw) { load_meta(cw, m0); fetch(m0, nxt); }
    if (cw1 < ncw) load_meta(cw1, m1);
    settle(m0, nxt, cur);
    while (cw < ncw) {
        const int cw2 = cw1 < ncw ? advance() : ncw;     // (a new ticket: an atomic, waited for before anything else goes out)
        if constexpr (ablated(22)) { if (prev_cw >= 0 && lane == 0) out_nit[prev_cw] = prev_nit; }
        else if (prev_cw >= 0) emit();
        if (cw1 < ncw) fetch(m1, nxt);
        if (cw2 < ncw) load_meta(cw2, m2);
        const uint32_t out_off = m0.out_off;
        const int k_used = m0.k_used;
        // Variables are numbered in ON-AIR order here (variable i of the tables == the reference's variable permute[i],
        // ldpc_dec.cc:64-66: y[permute[i]] = -llr[i]): lane l holds the soft bits of variables l, l + 64, ... as they
        // arrived, and nothing has to be scattered.  awgn_bp::compute_init_estimate: 1 where y = -llr < 0.
        unsigned long long X[NV];
#pragma unroll
        for (int p = 0; p < NV; p++) {
            const float y = -1 * cur[p];
            X[p] = __ballot(y < 0);                       // (positions >= n hold the shortened value: 0)
        }
        int nit = 0;
        if (!syndrome_ok(X)) {
            bool wild = false;                            // a channel exponent outside the tame range (or NaN)
#pragma unroll
            for (int p = 0; p < NV; p++) {
                const int v = lane + p * kWave;
                const float y = -1 * cur[p];
                float arg = (-1 * 2 * y) / s2;
                // written as comparisons: NaN (no ordering) takes the last branch -- the compiler may turn an fmin/fmax clamp
                // into a median instruction whose NaN result differs
                if (FAST) arg = arg > kFastLlrMax ? kFastLlrMax : (arg >= -kFastLlrMax ? arg : -kFastLlrMax);
                if (v < t.n) {
                    if constexpr (!ablated(25)) wild |= !(fabsf(arg) <= kTameArgMax);
                    const double lr = ablated(24) ? 1.0 + double(arg) : dm_exp64(double(arg));
                    rx[v] = lr;
                    if constexpr (!FAST) {
                        // spa_initialize + the first half of the first update_chks: Q_e = channel ratio of the edge's variable
                        // for every edge e of v, so t_e = 2/(1+Q_e) - 1 is ONE value per variable -- formed here once (the
                        // same IEEE operations on the same operands as the per-edge form: same bits) and written to the
                        // variable's message slots, where the first check pass finds it: 300 divisions instead of 888
                        const double tv = 2.0 / (1 + lr) - 1;
                        const unsigned long long packed = *reinterpret_cast<const unsigned long long*>(var_edge + v * 4);
                        unsigned char* msg_b = reinterpret_cast<unsigned char*>(msg);
#pragma unroll
                        for (int i = 0; i < MAXC; i++) *reinterpret_cast<double*>(msg_b + int((packed >> (16 * i)) & 0xFFFFull)) = tv;
                    }
                }
            }
            if constexpr (!FAST && !ROWS_FULL) {
                // a padded row slot (rows lighter than MAXR) belongs to no variable: its factor is exactly 1
                for (int c = lane; c < t.m; c += kWave)
                    for (int j = int(chk_deg[c]); j < MAXR; j++) msg[j * t.m + c] = 1.0;
            }
            if (lane == 0) rx[t.n] = 0.0;
            wave_sync();
            // (fast arithmetic: spa_initialize is folded into the first check pass, which reads rx through the row's
            // variable indices instead of the message slots)
            if (lane == 0) msg[kOneSlot] = 1.0;   // (a padded column entry named this slot: written above, put right here)
            wave_sync();
            auto iterate = [&](auto tame) {
                constexpr bool TAME = decltype(tame)::value;
                while (nit < max_it) {
                    nit += 1;
                    if (FAST) {
                        if (nit == 1) check_phase_fast(std::true_type{});
                        else check_phase_fast(std::false_type{});
                    } else {
                        if (nit == 1) check_phase(std::true_type{}, tame);
                        else check_phase(std::false_type{}, tame);
                    }
                    wave_sync();
                    // ---- update_vars + decision: lane == variable node.  var_edge holds BYTE offsets of a variable's
                    // message slots; a padded entry names the slot kept at 1.0 and is written like the others -- every
                    // lane's reads of a pass precede its writes (one wave, program order), and the slot is set back to
                    // 1.0 behind the writes of each pass (one uniform store instead of a compare + select per entry)
                    unsigned char* msg_b = reinterpret_cast<unsigned char*>(msg);
                    // The variable-to-check messages Q = sum / R of an iteration are read by the NEXT iteration only.  When
                    // the decisions of this iteration already satisfy every check -- or the iteration limit is reached -- they
                    // are never read: the exact arithmetic therefore forms the column products and the decisions first
                    // (DECIDE), runs the syndrome check, and divides only if the codeword goes on (DIVIDE: the products
                    // are formed again from the same operands -- 3 LDS reads and 3 multiplications per variable against 3
                    // divisions saved on a codeword's last iteration).  Codewords that have not converged after
                    // `split_iterations` iterations rarely stop at the next one: from there on the two halves run fused, as one
                    // pass (no second read), except in the last allowed iteration.  Every message that IS read is the
                    // reference's: same operations, same operands, same order.
                    const bool last_allowed = nit == max_it;
                    auto var_pass = [&](auto decide, auto divide) {
                        constexpr bool DECIDE = decltype(decide)::value, DIVIDE = decltype(divide)::value;
                        // (unrolled over the NV passes: the decisions of pass p are the compare's own result, word X[p])
#pragma unroll
                        for (int p = 0; p < NV; p++) {
                            const int v = lane + p * kWave;
                            bool one = false;
                            if (v < t.n) {
                                const unsigned long long packed = *reinterpret_cast<const unsigned long long*>(var_edge + v * 4);
                                int ee[MAXC];
                                double rr[MAXC];
                                if (FAST) {
                                    // Q_i = (rx R_0 .. R_{i-1}) (R_{i+1} .. R_{d-1}): prefix x suffix, no division
                                    double pre[MAXC + 1], suf[MAXC];
                                    pre[0] = rx[v];
#pragma unroll
                                    for (int i = 0; i < MAXC; i++) {
                                        ee[i] = int((packed >> (16 * i)) & 0xFFFFull);
                                        rr[i] = *reinterpret_cast<const double*>(msg_b + ee[i]);
                                        pre[i + 1] = pre[i] * rr[i];
                                    }
                                    if constexpr (DIVIDE) {
                                        suf[MAXC - 1] = 1.0;
#pragma unroll
                                        for (int i = MAXC - 2; i >= 0; i--) suf[i] = suf[i + 1] * rr[i + 1];
#pragma unroll
                                        for (int i = 0; i < MAXC; i++) *reinterpret_cast<double*>(msg_b + ee[i]) = pre[i] * suf[i];
                                    }
                                    one = pre[MAXC] > 1;
                                } else {
                                    double sum = rx[v];
#pragma unroll
                                    for (int i = 0; i < MAXC; i++) {
                                        ee[i] = int((packed >> (16 * i)) & 0xFFFFull);
                                        rr[i] = *reinterpret_cast<const double*>(msg_b + ee[i]);
                                        sum = sum * rr[i];
                                    }
                                    if constexpr (DIVIDE) {
#pragma unroll
                                        for (int i = 0; i < MAXC; i++) *reinterpret_cast<double*>(msg_b + ee[i]) = div_exact<TAME>(sum, rr[i]);
                                    }
                                    one = sum > 1;
                                }
                                if constexpr (DIVIDE) msg[kOneSlot] = 1.0;
                            }
                            if constexpr (DECIDE) X[p] = __ballot(one);
                        }
                    };
                    if (last_allowed || (!FAST && nit <= split_iterations)) {
                        var_pass(std::true_type{}, std::false_type{});
                        if (syndrome_ok(X) || last_allowed) break;    // (the check reads the decision words and the row masks only)
                        var_pass(std::false_type{}, std::true_type{});
                        wave_sync();
                    } else {
                        var_pass(std::true_type{}, std::true_type{});
                        wave_sync();
                        if (syndrome_ok(X)) break;
                    }
                }
            };
            // exact arithmetic: the cheaper division form when every channel exponent of this codeword is tame (see
            // div_exact); the choice is wave-uniform.  The fast arithmetic has no per-edge divisions to save.
            if constexpr (ablated(26)) { /* no iteration: what a codeword costs that fails its first syndrome check, without them */ }
            else if (FAST || __ballot(wild) != 0ull) iterate(std::false_type{});
            else iterate(std::true_type{});
        }
        // cldpc::get_systematic_bits: data bit i is variable permute[rank + i], i.e. on-air position rank + i -- the
        // output is the decision bitmap shifted down by `rank` bits.  The bitmap goes through the wave's LDS slice as
        // NV + 1 64-bit words (the last one zero); every output lane reads the dwords its bits straddle.
        {
            unsigned long long word = 0;
#pragma unroll
            for (int p = 0; p < NV; p++) word = lane == p ? X[p] : word;
            if (lane <= NV) reinterpret_cast<unsigned long long*>(xw)[lane] = word;
        }
        wave_sync();
        // (the stores themselves: `emit`, at the top of the next pass)
        prev_cw = cw; prev_nit = nit; prev_out_off = out_off; prev_k_used = k_used;
        settle(m1, nxt, cur);
        m0 = m1; m1 = m2;
        cw = cw1; cw1 = cw2;
    }
    if (prev_cw >= 0) emit();
}

} // namespace

LdpcDecoder::~LdpcDecoder()
{
    if (blob_) (void)hipFree(blob_);
}

size_t LdpcDecoder::lds_bytes() const
{
    return size_t(code.n_edges) * 8 + size_t(code.n) * 8 + size_t((code.n + 7) / 8) * 8;
}

int LdpcDecoder::upload()
{
    if (code.n <= 0 || code.n_edges <= 0) { set_last_error("ldpc: empty code"); return DTL_ERR_INVALID_ARG; }
    if (size_t(code.n) * 4 > size_t(code.n_edges) * 8) { set_last_error("ldpc: n too large vs edges"); return DTL_ERR_UNSUPPORTED; }
    // one allocation, 16-byte aligned sections
    std::vector<uint8_t> host;
    auto put = [&](const void* p, size_t bytes) {
        size_t off = (host.size() + 15) & ~size_t(15);
        host.resize(off + bytes);
        std::memcpy(host.data() + off, p, bytes);
        return off;
    };
    auto to16 = [](const std::vector<int>& v) { return std::vector<uint16_t>(v.begin(), v.end()); };
    std::vector<uint16_t> col_edge = to16(code.col_edge), row_ptr = to16(code.row_ptr),
                          edge_var = to16(code.edge_var), inv_perm(code.n), sys_src(code.k);
    for (int i = 0; i < code.n; i++) inv_perm[code.permute[i]] = uint16_t(i);
    for (int i = 0; i < code.k; i++) sys_src[i] = uint16_t(code.permute[i + code.rank]);
    size_t o_chk = put(code.chk_sched.data(), code.chk_sched.size() * sizeof(ChkSlot));
    size_t o_var = put(code.var_sched.data(), code.var_sched.size() * sizeof(VarSlot));
    size_t o_ce = put(col_edge.data(), col_edge.size() * 2);
    size_t o_rp = put(row_ptr.data(), row_ptr.size() * 2);
    size_t o_ev = put(edge_var.data(), edge_var.size() * 2);
    size_t o_ip = put(inv_perm.data(), inv_perm.size() * 2);
    size_t o_ss = put(sys_src.data(), sys_src.size() * 2);
    const std::vector<unsigned int> zeros(size_t(kNpQueueRing) * kNpQueues * kNpQueueStride, 0u);
    size_t o_q = put(zeros.data(), zeros.size() * sizeof(unsigned int));
    // fixed-stride tables of the node-parallel kernel (see the kernel's comment)
    const bool small_code = code.max_row_deg <= 6 && code.max_col_deg <= 3 && code.n <= 320; // kernel <6,3,5>
    const int maxr = small_code ? 6 : 8;
    const int n_slots = code.m * maxr;
    std::vector<uint16_t> chk_var(size_t(n_slots), uint16_t(code.n));          // pad -> dummy variable n
    std::vector<uint8_t> chk_deg(size_t(code.m), 0);
    std::vector<int> slot_of_edge(size_t(code.n_edges), 0);                    // check-major edge -> padded slot
    bool rows_full = true;
    for (int c = 0; c < code.m; c++) {
        const int deg = code.row_ptr[c + 1] - code.row_ptr[c];
        chk_deg[c] = uint8_t(std::min(deg, 255));
        if (deg != maxr) rows_full = false;
        for (int j = 0; j < deg && j < maxr; j++) {
            chk_var[size_t(c) * maxr + j] = inv_perm[code.edge_var[code.row_ptr[c] + j]];   // on-air numbering
            slot_of_edge[code.row_ptr[c] + j] = j * code.m + c;   // entry-major message layout
        }
    }
    // BYTE offsets of the message slots (8 bytes each; n_slots * 8 < 65536 is checked below)
    std::vector<uint16_t> var_edge(size_t(code.n) * 4, uint16_t(n_slots * 8)); // pad -> the slot kept at 1.0
    for (int v = 0; v < code.n; v++)
        for (int i = code.col_ptr[v]; i < code.col_ptr[v + 1] && i - code.col_ptr[v] < 4; i++)
            var_edge[size_t(inv_perm[v]) * 4 + (i - code.col_ptr[v])] = uint16_t(slot_of_edge[code.col_edge[i]] * 8);
    size_t o_cv = put(chk_var.data(), chk_var.size() * 2);
    size_t o_ve = put(var_edge.data(), var_edge.size() * 2);
    size_t o_cd = put(chk_deg.data(), chk_deg.size());
    // row bitmaps of H in on-air numbering: word p of check c (at [c * stride + p], stride = NV | 1) has bit l set iff
    // variable l + 64 p is in row c
    const int nv_words = small_code ? 5 : 8, nv_stride = nv_words | 1;
    std::vector<unsigned long long> chk_bits(size_t(code.m) * nv_stride, 0ull);
    for (int c = 0; c < code.m; c++)
        for (int e = code.row_ptr[c]; e < code.row_ptr[c + 1]; e++) {
            const int v = inv_perm[code.edge_var[e]];
            if (v < 64 * nv_words) chk_bits[size_t(c) * nv_stride + (v >> 6)] ^= 1ull << (v & 63);
        }
    size_t o_cb = put(chk_bits.data(), chk_bits.size() * 8);
    if (blob_) { (void)hipFree(blob_); blob_ = nullptr; }
    DTL_HIP_CHECK(hipMalloc(&blob_, host.size()));
    DTL_HIP_CHECK(hipMemcpy(blob_, host.data(), host.size(), hipMemcpyHostToDevice));
    auto* b = static_cast<uint8_t*>(blob_);
    t_.n = code.n; t_.m = code.m; t_.k = code.k; t_.rank = code.rank; t_.n_edges = code.n_edges;
    t_.n_chk_pass = code.n_chk_pass; t_.n_var_pass = code.n_var_pass;
    t_.chk = reinterpret_cast<const ChkSlot*>(b + o_chk);
    t_.var = reinterpret_cast<const VarSlot*>(b + o_var);
    t_.col_edge = reinterpret_cast<const uint16_t*>(b + o_ce);
    t_.row_ptr = reinterpret_cast<const uint16_t*>(b + o_rp);
    t_.edge_var = reinterpret_cast<const uint16_t*>(b + o_ev);
    t_.inv_perm = reinterpret_cast<const uint16_t*>(b + o_ip);
    t_.sys_src = reinterpret_cast<const uint16_t*>(b + o_ss);
    np_chk_var_ = reinterpret_cast<const uint16_t*>(b + o_cv);
    np_var_edge_ = reinterpret_cast<const uint16_t*>(b + o_ve);
    np_chk_deg_ = b + o_cd;
    np_chk_bits_ = reinterpret_cast<const unsigned long long*>(b + o_cb);
    np_rows_full_ = rows_full;
    np_maxr_ = maxr;
    np_queue_ = reinterpret_cast<unsigned int*>(b + o_q);
    // LDS plan of the node-parallel kernel: shared index tables + 16 per-wave slices
    {
        auto al = [](size_t v) { return (v + 15) & ~size_t(15); };
        size_t off = 0;
        np_sh_.chk_var = int(off); off = al(off + size_t(n_slots) * 2);
        np_sh_.var_edge = int(off); off = al(off + size_t(code.n) * 4 * 2);
        np_sh_.chk_deg = int(off); off = al(off + size_t(code.m));
        np_sh_.chk_bits = int(off); off = al(off + size_t(code.m) * nv_stride * 8);
        np_sh_.wave_base = int(off);
        // per-wave slice: padded message slots + 2, channel LRs + 1 (their first 9 words double as the decision bitmap
        // on its way out: n >= 9 is checked below)
        np_sh_.wave_stride = int(al(size_t(n_slots + 2) * 8 + size_t(code.n + 1) * 8));
        np_lds_ = off + size_t(kNpWaves) * np_sh_.wave_stride;
        np_ok_ = code.max_row_deg <= 8 && code.max_col_deg <= 4 && code.n <= 512 && code.n >= 9 && (n_slots + 2) * 8 < 65535 &&
                 np_lds_ <= 160 * 1024;
        if (np_ok_) {
            for (auto kern : { reinterpret_cast<const void*>(ldpc_bp_np_kernel<6, 3, 5, true, false>),
                               reinterpret_cast<const void*>(ldpc_bp_np_kernel<6, 3, 5, false, false>),
                               reinterpret_cast<const void*>(ldpc_bp_np_kernel<8, 4, 8, true, false>),
                               reinterpret_cast<const void*>(ldpc_bp_np_kernel<8, 4, 8, false, false>),
                               reinterpret_cast<const void*>(ldpc_bp_np_kernel<6, 3, 5, true, true>),
                               reinterpret_cast<const void*>(ldpc_bp_np_kernel<6, 3, 5, false, true>),
                               reinterpret_cast<const void*>(ldpc_bp_np_kernel<8, 4, 8, true, true>),
                               reinterpret_cast<const void*>(ldpc_bp_np_kernel<8, 4, 8, false, true>) })
                DTL_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, int(np_lds_)));
        }
    }
    return DTL_OK;
}

int LdpcDecoder::decode(const float* d_llr, const CwDesc* d_desc, size_t ncw, uint8_t* d_bits,
                        int32_t* d_nit, hipStream_t stream, bool packed) const
{
    if (!blob_) { set_last_error("ldpc: tables not uploaded"); return DTL_ERR_INVALID_ARG; }
    if (ncw == 0) return DTL_OK;
    if (ncw > size_t(0x7fffffff)) { set_last_error("ldpc: batch too large"); return DTL_ERR_INVALID_ARG; }
    if (!d_llr || !d_bits || !d_nit) { set_last_error("ldpc: null buffer"); return DTL_ERR_INVALID_ARG; }
    if (!d_desc && ncw * size_t(code.n) > size_t(0xffffffffu)) {
        set_last_error("ldpc: dense batch exceeds 32-bit soft-bit offsets");
        return DTL_ERR_INVALID_ARG;
    }
    if (np_ok_) {
        const int grid = int(std::min<size_t>((ncw + kNpWaves - 1) / kNpWaves, 256));
        auto kern = np_maxr_ == 6 ? (np_rows_full_ ? ldpc_bp_np_kernel<6, 3, 5, true, false> : ldpc_bp_np_kernel<6, 3, 5, false, false>)
                                  : (np_rows_full_ ? ldpc_bp_np_kernel<8, 4, 8, true, false> : ldpc_bp_np_kernel<8, 4, 8, false, false>);
        if (arith == DTL_ARITH_FAST)
            kern = np_maxr_ == 6 ? (np_rows_full_ ? ldpc_bp_np_kernel<6, 3, 5, true, true> : ldpc_bp_np_kernel<6, 3, 5, false, true>)
                                 : (np_rows_full_ ? ldpc_bp_np_kernel<8, 4, 8, true, true> : ldpc_bp_np_kernel<8, 4, 8, false, true>);
        // every wave stops at its first out-of-range ticket: the counter ends below n_chunks + #waves.
        // Large batches take 16 codewords per ticket (keeps the same-address atomics off the critical
        // path); small ones shrink the ticket so that all 4096 waves get work and the tail stays short (at 25 k
        // codewords, C2, two per ticket measured 0.36 ms against 0.44 ms for one or four).
        // (DTL_LDPC_CHUNK: tuning experiments, tools/ldpc_probe.py)
        static const int chunk_cap = [] { const char* e = std::getenv("DTL_LDPC_CHUNK"); return e ? std::max(1, atoi(e)) : kNpChunk; }();
        const int chunk = int(std::max<size_t>(1, std::min<size_t>(size_t(chunk_cap), ncw / (size_t(grid) * kNpWaves * 3))));
        // every launch takes its own ticket counter from a ring: two decodes of one decoder on different streams
        // (legal: decode() is const and takes the caller's stream) never share one
        unsigned int* queue = np_queue_ + size_t(np_launch_.fetch_add(1, std::memory_order_relaxed) % kNpQueueRing) * kNpQueues * kNpQueueStride;
        DTL_HIP_CHECK(hipMemsetAsync(queue, 0, sizeof(unsigned int) * kNpQueues * kNpQueueStride, stream));
        // iterations whose variable pass forms the decisions first and divides only if the codeword goes on (see the kernel);
        // (DTL_LDPC_SPLIT_IT: tuning experiments)
        static const int split_it = [] { const char* e = std::getenv("DTL_LDPC_SPLIT_IT"); return e ? atoi(e) : kNpSplitIterations; }();
        // taking tickets from the other counters when the home counter has run out pays once a wave has many codewords to
        // balance (DTL_LDPC_STEAL_MIN: tuning experiments)
        static const size_t steal_min = [] { const char* e = std::getenv("DTL_LDPC_STEAL_MIN"); return e ? size_t(atoll(e)) : size_t(1) << 17; }();
        const int steal = (ncw >= steal_min || grid < kNpQueues) ? 1 : 0;   // (fewer workgroups than counters: some counter has no wave of its own)
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kNpWaves * kWave), np_lds_, stream, t_, np_sh_, np_chk_var_, np_var_edge_, np_chk_deg_, np_chk_bits_,
                           d_llr, d_desc, int(ncw), max_iterations, sigma, d_bits, d_nit, queue, chunk, packed ? 1 : 0, split_it, steal);
        DTL_HIP_CHECK(hipGetLastError());
        return DTL_OK;
    }
    if (packed) { set_last_error("ldpc: packed output is implemented by the node-parallel kernel only"); return DTL_ERR_UNSUPPORTED; }
    if (arith == DTL_ARITH_FAST) {
        set_last_error("ldpc: the fast arithmetic is implemented by the node-parallel kernel only (row weight <= 8, column weight <= 4, n <= 512)");
        return DTL_ERR_UNSUPPORTED;
    }
    const size_t lds = lds_bytes();
    const int waves_per_cu = int(std::min<size_t>(16, (160 * 1024) / lds));
    int grid = int(std::min<size_t>(ncw, size_t(256) * waves_per_cu));
    hipLaunchKernelGGL(ldpc_bp_kernel, dim3(grid), dim3(kWave), lds, stream, t_, d_llr, d_desc, int(ncw),
                       max_iterations, sigma, d_bits, d_nit);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

} // namespace dtl
