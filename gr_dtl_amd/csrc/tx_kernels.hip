// tx_kernels.hip -- HIP kernels (gfx950) of the batched transmitter.
//
//   tx_encode_kernel  one wave per transport block: CRC-32 of the user bytes, LSB-first bit string,
//                     tb_encoder's even split into codewords, LDPC encoding as a GF(2) matrix-vector
//                     product on bit-packed words (popcount), coded bits [check | data] per codeword.
//   tx_frame_kernel   one N-thread workgroup per frame: coded bits -> symbols (repack LSB first) at the
//                     positions the framing state machine chose, header bits -> BPSK, carrier allocation
//                     with pilots, inverse FFT (same LDS radix-2 butterflies and twiddle table as the
//                     receive kernels and the CPU oracle, conjugated), cyclic prefix, coalesced store.
#include "tx_pipeline.h"

#pragma clang fp contract(off)

namespace dtl {

namespace {

struct TxCodeSet {
    TxCodeTables c[8];
};

// twiddle product: one fused multiply-add per component, as in the receive kernels and the oracle's FFT
__device__ __forceinline__ cf32 cmul(cf32 a, cf32 w)
{
    return { __builtin_fmaf(a.re, w.re, -(a.im * w.im)), __builtin_fmaf(a.re, w.im, a.im * w.re) };
}

// ------------------------------------------------------------------------------------------
// ofdm_adaptive_fec_frame_bvb_impl.cc:397-408 (user bytes + CRC-32, unpacked LSB first),
// tb_encoder::encode (tb_encoder.cc:32-74), ldpc_enc::encode (ldpc_enc.cc:58-68: [parity | data]).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tx_encode_kernel(TxCodeSet codes, const TxTb* __restrict__ tbs, int n_tbs,
                                                      const uint8_t* __restrict__ user, uint8_t* __restrict__ coded,
                                                      int max_bytes, int max_cws)
{
    extern __shared__ unsigned char tx_smem[];
    uint32_t* crc_tab = reinterpret_cast<uint32_t*>(tx_smem);             // 256
    unsigned long long* dwords = reinterpret_cast<unsigned long long*>(crc_tab + 256); // 8 packed data words
    int* ds = reinterpret_cast<int*>(dwords + 8);                          // max_cws + 1 data-bit prefix
    uint8_t* buf = reinterpret_cast<uint8_t*>(ds + max_cws + 1);           // user bytes + CRC
    const int lane = threadIdx.x;
    if (blockIdx.x >= n_tbs) return;
    const TxTb tb = tbs[blockIdx.x];
    for (int i = lane; i < 256; i += 64) {
        uint32_t c = uint32_t(i);
        for (int b = 0; b < 8; b++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        crc_tab[i] = c;
    }
    for (uint32_t i = lane; i < tb.user_len; i += 64) buf[i] = user[size_t(tb.user_off) + i];
    __syncthreads();
    if (lane == 0) {
        uint32_t crc = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < tb.user_len; i++) crc = crc_tab[(crc ^ buf[i]) & 0xFFu] ^ (crc >> 8);
        crc ^= 0xFFFFFFFFu;
        for (int i = 0; i < 4; i++) buf[tb.user_len + i] = uint8_t(crc >> (8 * i)); // little endian
        // even split of the data bits over the codewords (tb_encoder.cc:41-47): k_i = ceil(rem / (ncws - i))
        int rem = int(tb.user_len + 4) * 8, pos = 0;
        ds[0] = 0;
        for (int i = 0; i < tb.ncws; i++) {
            const int kn = (rem + (tb.ncws - i) - 1) / (tb.ncws - i);
            pos += kn; rem -= kn;
            ds[i + 1] = pos;
        }
    }
    __syncthreads();
    const int n_bits = int(tb.user_len + 4) * 8;
    auto bit = [&](int i) -> unsigned { return (buf[i >> 3] >> (i & 7)) & 1u; };
    uint8_t* out = coded + size_t(tb.coded_off);
    if (tb.ncws == 0) { // no FEC: the raw bit string, zero padded to the frame's byte count
        for (int i = lane; i < int(tb.coded_len); i += 64) out[i] = i < n_bits ? uint8_t(bit(i)) : uint8_t(0);
        return;
    }
    const TxCodeTables code = codes.c[tb.code];
    const int ncheck = code.n - code.k;
    for (int c = 0; c < tb.ncws; c++) {
        const int d0 = ds[c], kn = ds[c + 1] - d0;
        if (lane < code.words) {
            unsigned long long w = 0;
            for (int j = 0; j < 64; j++) {
                const int p = lane * 64 + j;
                if (p < kn) w |= static_cast<unsigned long long>(bit(d0 + p)) << j;
            }
            dwords[lane] = w;
        }
        __syncthreads();
        uint8_t* cw = out + size_t(c) * ncheck + d0;
        for (int i = lane; i < code.rank; i += 64) {
            int acc = 0;
            for (int w = 0; w < code.words; w++) acc += __popcll(code.parity[size_t(i) * code.words + w] & dwords[w]);
            cw[i] = uint8_t(acc & 1);
        }
        for (int j = lane; j < kn; j += 64) cw[ncheck + j] = uint8_t(bit(d0 + j));
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// chunks_to_symbols / repack(1, bps) LSB first, ofdm_carrier_allocator_cvc, fft_vcc(inverse, shift),
// ofdm_cyclic_prefixer (python/dtl/ofdm_transmitter.py:118-189).
// ------------------------------------------------------------------------------------------
__global__ void tx_frame_kernel(TxTables t, const TxFrame* __restrict__ frames, const TxSeg* __restrict__ segs,
                                const uint8_t* __restrict__ coded, const uint8_t* __restrict__ hdr_bits,
                                cf32* __restrict__ iq, int64_t frame_stride, uint8_t* __restrict__ syms_out)
{
    extern __shared__ unsigned char tx_smem[];
    const int N = t.fft_len, P = N + t.cp_len, h = t.header_len, L = t.frame_length, cap = t.frame_capacity;
    const int tid = threadIdx.x;
    cf32* bufA = reinterpret_cast<cf32*>(tx_smem);
    cf32* bufB = bufA + N;
    cf32* tw = bufB + N;                        // N/2
    uint8_t* syms = reinterpret_cast<uint8_t*>(tw + N / 2); // cap
    uint8_t* hb = syms + ((cap + 15) & ~15);    // hdr_bits
    if (tid < N / 2) tw[tid] = t.twiddle[tid];
    const TxFrame fr = frames[blockIdx.x];
    const int bps = fr.cnst == 4 ? 4 : fr.cnst;
    for (int i = tid; i < cap; i += blockDim.x) syms[i] = 0;
    for (int i = tid; i < t.hdr_bits; i += blockDim.x) hb[i] = hdr_bits[size_t(blockIdx.x) * t.hdr_bits + i];
    for (int sgi = 0; sgi < fr.seg_count; sgi++) {
        const TxSeg sg = segs[fr.seg_first + sgi];
        __syncthreads();
        for (int i = sg.sym_pos + tid; i < cap; i += blockDim.x) syms[i] = 0; // cur[wpos:] = 0
        __syncthreads();
        const int nsym = (int(sg.nbits) + bps - 1) / bps;
        for (int j = tid; j < nsym; j += blockDim.x) {
            if (sg.sym_pos + j >= cap) break;
            unsigned v = 0;
            for (int b = 0; b < bps; b++) {
                const int p = j * bps + b;
                if (p < int(sg.nbits)) v |= unsigned(coded[size_t(sg.coded_off) + p] & 1u) << b;
            }
            syms[sg.sym_pos + j] = uint8_t(v);
        }
    }
    __syncthreads();
    if (syms_out) for (int i = tid; i < cap; i += blockDim.x) syms_out[size_t(blockIdx.x) * cap + i] = syms[i];

    const cf32* pts = t.cnst_pts + fr.cnst * 16;
    const cf32* bpsk = t.cnst_pts + 1 * 16;
    cf32* out = iq + size_t(blockIdx.x) * size_t(frame_stride);
    const int br = t.bitrev[tid];
    const int pos = (tid + N / 2) & (N - 1);    // ifftshift: FFT input k comes from shifted position (k + N/2) % N
    const int n_sym = 2 + h + L;
    int hdr_pos = 0, pay_pos = 0;               // data symbols consumed before OFDM symbol i (uniform)
    auto freq_value = [&](int s, int hp, int pp) -> cf32 {
        if (s == 0) return t.sync1[pos];
        if (s == 1) return t.sync2[pos];
        const int i = s - 2;
        cf32 v = { 0.0f, 0.0f };
        const int r = t.occ_rank[(i % t.n_occ_sets) * N + pos];
        if (r >= 0) v = (i < h) ? bpsk[hb[hp + r] & 1u] : pts[syms[pp + r]];
        if (t.n_pilot_sets) {
            const int pr = t.pil_rank[(i % t.n_pilot_sets) * N + pos];
            if (pr >= 0) v = t.pilot_syms[(i % t.n_pilot_sym_sets) * t.n_pilots + pr];
        }
        return v;
    };
    auto advance = [&](int s) { // account for the data symbols of OFDM symbol s
        if (s < 2) return;
        const int i = s - 2, len = t.occ_len[i % t.n_occ_sets];
        if (i < h) hdr_pos += len; else pay_pos += len;
    };
    const int hN = N / 2, logN = __ffs(N) - 1;
    for (int s0 = 0; s0 < n_sym; s0 += 2) {
        const bool two = s0 + 1 < n_sym;
        bufA[br] = freq_value(s0, hdr_pos, pay_pos);
        advance(s0);
        if (two) { bufB[br] = freq_value(s0 + 1, hdr_pos, pay_pos); advance(s0 + 1); }
        __syncthreads();
        const bool upper = tid >= hN;
        cf32* buf = upper ? bufB : bufA;
        const int bt = upper ? tid - hN : tid;
        const bool active = !upper || two;
        for (int st = 0; st < logN; st++) { // butterflies of span 2^(st+1); shifts, not divisions
            const int half = 1 << st;
            if (active) {
                const int grp = bt >> st, k = bt & (half - 1);
                const int a = (grp << (st + 1)) + k, b = a + half;
                const cf32 w = tw[k << (logN - st - 1)];
                const cf32 u = buf[a];
                const cf32 vv = cmul(buf[b], w);
                buf[a] = { u.re + vv.re, u.im + vv.im };
                buf[b] = { u.re - vv.re, u.im - vv.im };
            }
            __syncthreads();
        }
        for (int u = 0; u < (two ? 2 : 1); u++) {
            const cf32 v = (u ? bufB : bufA)[tid];
            cf32* o = out + size_t(s0 + u) * P;
            o[t.cp_len + tid] = v;
            if (tid >= N - t.cp_len) o[tid - (N - t.cp_len)] = v;
        }
        __syncthreads();
    }
}

} // namespace

int launch_tx_encode(const TxCodeTables* codes_by_id, int n_codes, const TxTb* tbs, int n_tbs, int max_user_len,
                     int max_cws, const uint8_t* user, uint8_t* coded, hipStream_t st)
{
    if (n_tbs <= 0) return DTL_OK;
    if (n_codes > 8) { set_last_error("tx: at most 7 FEC codes"); return DTL_ERR_UNSUPPORTED; }
    TxCodeSet set{};
    int max_k = 0;
    for (int i = 0; i < n_codes; i++) { set.c[i] = codes_by_id[i]; max_k = std::max(max_k, codes_by_id[i].k); }
    if ((max_k + 63) / 64 > 8) { set_last_error("tx: code dimension above 512 not supported by the encoder kernel"); return DTL_ERR_UNSUPPORTED; }
    const int max_bytes = max_user_len + 4;
    const size_t lds = 256 * 4 + 8 * 8 + size_t(max_cws + 1) * 4 + size_t(max_bytes) + 8;
    if (lds > 160 * 1024) { set_last_error("tx: transport block does not fit the LDS"); return DTL_ERR_UNSUPPORTED; }
    static size_t attr_lds = 64 * 1024;
    if (lds > attr_lds) {
        DTL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tx_encode_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        attr_lds = lds;
    }
    hipLaunchKernelGGL(tx_encode_kernel, dim3(n_tbs), dim3(64), lds, st, set, tbs, n_tbs, user, coded, max_bytes, max_cws);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

int launch_tx_frames(const TxTables& t, const TxFrame* frames, int n_frames, const TxSeg* segs, const uint8_t* coded,
                     const uint8_t* hdr_bits, cf32* iq, int64_t frame_stride, uint8_t* syms_out, hipStream_t st)
{
    if (n_frames <= 0) return DTL_OK;
    const int N = t.fft_len;
    const size_t lds = size_t(2 * N + N / 2) * sizeof(cf32) + size_t((t.frame_capacity + 15) & ~15) + size_t(t.hdr_bits) + 16;
    if (lds > 160 * 1024) { set_last_error("tx: frame does not fit the LDS"); return DTL_ERR_UNSUPPORTED; }
    static size_t attr_lds = 64 * 1024;
    if (lds > attr_lds) {
        DTL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tx_frame_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        attr_lds = lds;
    }
    hipLaunchKernelGGL(tx_frame_kernel, dim3(n_frames), dim3(N), lds, st, t, frames, segs, coded, hdr_bits, iq, frame_stride,
                       syms_out);
    DTL_HIP_CHECK(hipGetLastError());
    return DTL_OK;
}

} // namespace dtl
