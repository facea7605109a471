// tx_pipeline.h -- batched adaptive-OFDM transmitter: the generator of on-air frames.
//
// Mirrors the TX graph of python/dtl/ofdm_transmitter.py:63-189 for one PDU and a constant MCS:
// ofdm_adaptive_fec_frame_bvb (lib/dtl/ofdm_adaptive_fec_frame_bvb_impl.cc:284-565) with tb_encoder
// (lib/dtl/tb_encoder.cc:32-74) and ldpc_enc (lib/dtl/ldpc_enc.cc:58-68), or ofdm_adaptive_frame_bb
// (lib/dtl/ofdm_adaptive_frame_bb_impl.cc:131-163); ofdm_adaptive_packet_header formatter
// (lib/dtl/ofdm_adaptive_packet_header.cc:141-199); chunks_to_symbols, ofdm_carrier_allocator_cvc,
// fft_vcc(inverse, shift), ofdm_cyclic_prefixer.
// The sequential framing state machine runs on the host on LENGTHS only (which bits of which transport
// block land where in which frame); CRC-32, LDPC encoding, symbol mapping, carrier allocation, IFFT and
// cyclic prefix run in two kernels (tx_kernels.hip).
#pragma once

#include "rx_pipeline.h"

namespace dtl {

struct TxFrameInfo { // mirrors dtl_tx_frame_info
    int32_t payload_len, frame_no, cnst, fb_cnst, tb_no, fec_fb, tb_offset, fec, tb_payload, pad;
};

// one transport block (FEC) or one frame's byte string (no FEC) to encode
struct TxTb {
    uint32_t user_off, user_len;   // user bytes of this block
    uint32_t coded_off;            // first coded bit (1 bit per byte) in the coded stream
    uint32_t coded_len;            // coded bits
    int32_t ncws;                  // codewords (0: no FEC -- raw bits of user + CRC-32, zero padded to coded_len)
    int32_t code;                  // FEC id
};

// a run of coded bits mapped to consecutive symbols of one frame (repack(1, bps) LSB first)
struct TxSeg {
    uint32_t coded_off, nbits;
    int32_t sym_pos;
    int32_t pad;
};

struct TxFrame {
    int32_t seg_first, seg_count;
    int32_t cnst, pad;
};

struct TxTables {
    int fft_len, cp_len, frame_length, header_len;
    int n_occ_sets, n_pilot_sets, n_pilots, n_pilot_sym_sets;
    int frame_capacity, hdr_bits;
    const cf32* twiddle;        // fft_len/2 : exp(+j 2 pi k / N)
    const uint16_t* bitrev;     // fft_len
    const int16_t* occ_rank;    // n_occ_sets x fft_len : index of the data carrier at a shifted position, -1 if none
    const int16_t* occ_len;     // n_occ_sets
    const int16_t* pil_rank;    // n_pilot_sets x fft_len : pilot index at a shifted position, -1 if none
    const cf32* pilot_syms;     // n_pilot_sym_sets x n_pilots
    const cf32* sync1;          // fft_len (shifted order, as configured)
    const cf32* sync2;
    const cf32* cnst_pts;       // 5 x 16
};

// LDPC generator part P (G = [I | P]) bit-packed for the encoder kernel
struct TxCodeTables {
    int n, k, rank, words;      // words = ceil(k / 64)
    const unsigned long long* parity; // rank x words
};

class Transmitter {
public:
    RxConfig cfg;               // the TX uses the carrier / pilot / sync word / FEC fields
    int init();
    // user bytes on the device; frames are written back to back (plus `gap` zero samples after each)
    int process(const uint8_t* d_user, int64_t n_user, int cnst, int fec, int first_frame_no, int first_tb_no,
                int gap, bool want_syms, hipStream_t stream);

    std::vector<TxFrameInfo> frames;
    int64_t n_samples = 0, n_tbs = 0;
    DevBuf d_iq, d_syms, d_hdr_bits;
    int frame_capacity() const { return capacity_; }
    int frame_samples() const { return (2 + cfg.header_len() + cfg.frame_length) * (cfg.fft_len + cfg.cp_len); }
    int hdr_bits() const { return cfg.header_bits(); }

    // framing state machine on lengths (host only): fills `frames` and the internal gather tables
    int plan(int64_t n_user, int cnst, int fec, int first_frame_no, int first_tb_no);

private:
    int upload_tables();
    bool uploaded_ = false;
    int capacity_ = 0;
    std::vector<LdpcCode> codes_; // index 0 unused ("no FEC")
    TxTables tab_{};
    std::vector<TxCodeTables> code_tab_;
    DevBuf tab_blob_;
    std::vector<TxTb> tbs_;
    std::vector<TxSeg> segs_;
    std::vector<TxFrame> tframes_;
    std::vector<uint8_t> hdr_bits_;
    uint64_t coded_total_ = 0;
    DevBuf d_tbs_, d_segs_, d_tframes_, d_coded_;
    HostBuf h_tbs_, h_segs_, h_tframes_, h_hdr_;
};

// kernel launchers (tx_kernels.hip)
int launch_tx_encode(const TxCodeTables* codes_by_id, int n_codes, const TxTb* tbs, int n_tbs, int max_user_len,
                     int max_cws, const uint8_t* user, uint8_t* coded, hipStream_t st);
int launch_tx_frames(const TxTables& t, const TxFrame* frames, int n_frames, const TxSeg* segs, const uint8_t* coded,
                     const uint8_t* hdr_bits, cf32* iq, int64_t frame_stride, uint8_t* syms_out, hipStream_t st);

// ofdm_adaptive_packet_header::header_formatter on the host (1 bit per byte)
void format_header_bits(const TxFrameInfo& f, bool has_fec, int header_len_bits, uint8_t* out);

} // namespace dtl
