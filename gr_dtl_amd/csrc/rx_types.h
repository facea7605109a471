// rx_types.h -- data structures shared by the host pipeline and the HIP kernels of the
// adaptive-OFDM receive path (python/dtl/ofdm_receiver.py:59-246 of the reference).
#pragma once

#include <cstdint>

namespace dtl {

struct cf32 {
    float re, im;
};

constexpr int kMaxBps = 4;
constexpr int kMaxHdrSyms = 4;

// LDS placement of element e of one FFT row (8-byte elements, 16 per 128-byte bank row).  The butterfly passes
// touch elements 4, 16, 64, ... apart and the bit-reversed load scatters consecutive lanes N/2, N/4, ... apart --
// all multiples of the bank row, i.e. the same banks.  XOR-ing the position inside a bank row with a linear
// function of the row number spreads every one of these access patterns (for 16 consecutive lanes) over all 16
// positions; rows stay rows, so no padding is needed.  Linear: swz(a ^ b) == swz(a) ^ swz(b).
__host__ __device__ inline int fft_lds_swizzle(int e)
{
    return e ^ (5 * ((e >> 4) & 3)) ^ ((e >> 6) & 3) ^ (((e >> 8) & 3) << 2);
}

// Read-only device tables derived from the configuration (one blob, see Receiver::upload_tables).
struct RxTables {
    int fft_len, cp_len, frame_length, header_len, has_fec;
    int n_occ_sets, max_occ, n_pilot_sets, n_pilots, n_pilot_sym_sets;
    int frame_capacity;        // QAM symbols per frame (python/dtl/ofdm_adaptive.py:1-6)
    int hdr_bits;              // header_len * len(occupied_carriers[0])
    int g_min, g_max;          // integer-CFO search range (even steps), ofdm_chanest_vcvc ctor
    float eq_alpha;
    int llr_dist_mode;
    const cf32* twiddle;       // fft_len : exp(-j 2 pi k / N) regrouped per butterfly pass (see fft_pass_twiddles)
    const uint16_t* bitrev;    // fft_len : bit-reversed index, already LDS-swizzled (fft_lds_swizzle)
    const int16_t* occ_idx;    // n_occ_sets x max_occ : shifted positions of the data carriers
    const int16_t* occ_len;    // n_occ_sets
    const int16_t* occ_rank;   // n_occ_sets x fft_len : index of the data carrier at a shifted position, -1 if none
    const uint8_t* occupied;   // fft_len : union of all data-carrier sets (shifted positions)
    const uint8_t* pilot_mask; // n_pilot_sets x fft_len
    const cf32* pilot_val;     // n_pilot_sym_sets x fft_len
    const cf32* corr_v;        // fft_len : sync2 / sync1 on sync1's non-zero carriers
    const uint16_t* sync_act;  // n_sync_act : the carriers with corr_v != 0, ascending
    int n_sync_act;
    const cf32* ref_sym;       // fft_len : sync word 2
    const cf32* cnst_pts;      // 5 x 16 constellation points (index = constellation_type_t)
    const double* recip_count; // frame_length * fft_len + 1 : RN(1 / k), the sample counts of the SNR estimator
};

// A stretch of the stream over which the fine-CFO oscillator runs at one held angle
// (sample_and_hold_ff + frequency_modulator_fc, ofdm_receiver.py:75,89).  The phase is kept in units of 2 pi / 2^32
// and wraps by itself: phase(i) = phase0 + inc (i - start + 1) mod 2^32, phase0 of the next segment = phase0 + inc x
// (next start - start) -- integer arithmetic, the same on the host, in the kernels and in the oracle (orc_cfo_mix_trig);
// the reference integrates a float and reads a 10-bit sine table (1e-4 rad), so any exact definition is ours to choose.
struct CfoSegment {
    int64_t start;    // first sample index at which the angle is held
    uint32_t phase0;  // oscillator phase just before `start`
    uint32_t inc;     // phase step per sample: osc_phase_step(sensitivity x held angle)
};
// radians per sample -> phase step (round to nearest, ties to even; anything that is not a finite, small number: 0)
inline uint32_t osc_phase_step(float radians)
{
    const double v = double(radians) * (4294967296.0 / (2.0 * 3.14159265358979323846));
    if (!(v > -9.0e15 && v < 9.0e15)) return 0u;
    return uint32_t(int64_t(__builtin_llrint(v)));
}

// One Schmidl-Cox trigger handed to the header kernel (phase A).
struct Candidate {
    int64_t trigger;  // index in the (delayed) stream the demux cuts at
    int32_t seg;      // index of the CfoSegment active at `trigger`
    int32_t pad;
};

// Raw result of phase A for one candidate (header_parser fields before the sequential
// "keep previous MCS" rule of lib/dtl/ofdm_adaptive_packet_header.cc:269-273).
struct HeaderRec {
    int32_t crc_ok;
    int32_t carr_offset;
    int32_t payload_len, frame_no, cnst, fb_cnst;
    int32_t tb_no, fec_fb, tb_offset, fec, tb_payload;
    int32_t pad;
};

// Extra arguments of the header kernel when it serves the per-block entry points dtl_chanest / dtl_header_decode
// (frequency-domain vectors in, see header_kernel<true>); all null / zero for the receive path.
struct HeaderStageIo {
    const cf32* freq_in = nullptr;       // per block: n_rows x N vectors, shifted (DC at N/2)
    int n_rows = 0, row0 = 0;            // rows supplied and the row of the (sync1, sync2, header...) stack the first one is
    const cf32* taps_in = nullptr;       // non-null: channel estimate supplied (skip ofdm_chanest_vcvc)
    const int32_t* carr_off_in = nullptr;
    cf32* taps_raw_out = nullptr;        // the estimator's taps ("ofdm_sync_chan_taps")
    int32_t* carr_off_out = nullptr;     // "ofdm_sync_carr_offset"
    int do_header = 0;                   // run the header equaliser, slicer and parser
};

// One accepted frame handed to the payload kernel (phase B).
struct FrameDesc {
    int64_t trigger;
    int32_t seg;
    int32_t cand;          // candidate index (taps are stored per candidate)
    int32_t cnst;          // constellation_type_t of the payload
    int32_t carr_offset;
    int32_t n_syms;        // packet_length tag: QAM symbols carried (0 => empty frame)
    uint32_t llr_off;      // where the frame's soft bits go in the call's soft-bit stream (FEC)
};

struct FrameStat {
    double snr_db;      // mpsk_snr_est_simple::snr()
    double noise_db;    // ::noise()
    double noise_lin;   // pow(10, noise_db / 10)  (equalizer.cc:281-287)
};

// One transport block to verify (lib/dtl/ofdm_adaptive_fec_decoder_impl.cc:161-181).
struct TbDesc {
    uint32_t bits_off;   // first decoded bit (1/byte) of the TB in the LDPC output buffer
    uint32_t n_bits;     // tb_payload_len (user bits + 32 CRC bits)
    uint32_t cw_first;   // first codeword index
    uint32_t n_cw;
};

// Transport block as a gather recipe: up to 4 source segments of the soft-bit stream (the receive
// buffer of tb_decoder.cc expressed as index ranges) + the even split of the payload over n_cw
// codewords (tb_decoder.cc:168-172).  Expanded on the device into one CwDesc per codeword.
struct TbGather {
    uint32_t seg_off[4];  // 0xFFFFFFFF = gap (soft bits never received)
    int32_t seg_len[4];
    uint32_t bits_off;    // first decoded bit of the TB in the LDPC output buffer
    uint32_t cw_first;    // global index of the TB's first codeword
    int32_t tb_payload;   // payload bits (user + CRC-32)
    int16_t n_cw, n_seg;
    int16_t ncheck, k;
    int32_t pad;
};

// One piece of a transport block whose soft bits are copied into a contiguous scratch run behind the call's soft-bit
// stream (blocks spread over more pieces than a TbGather / CwDesc can describe): the reference's RCV_BUF copy, made
// only for those blocks.
struct LlrCopy {
    uint32_t src;         // 0xFFFFFFFF = gap (never received): zeros
    uint32_t dst;
    int32_t len;
    int32_t pad;
};

struct TbResult {
    int32_t crc_ok;
    int32_t avg_it;
    int32_t user_bits;
    int32_t pad;
};

} // namespace dtl
