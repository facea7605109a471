/*
 * oracle.h -- CPU restatement ("oracle") of gr-dtl's adaptive-OFDM receive path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link,
 * import or call it -- and there only as the checker / reported CPU baseline.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * gr-dtl tree).  The half of the path that lives in GNU Radio 3.10 (not present in
 * the reference tree, not installable here) is restated from GR 3.10 semantics; those
 * functions are marked "GR3.10 (unverified restatement)".
 *
 * PARITY PINNING: the oracle is pinned against every known answer the reference's own
 * tests hold for this path (tests/golden/ JSON files, see tests/test_oracle_golden.py):
 * header bit vectors + CRC-16, LSB-first repack vector, 8PSK/BPSK metric constant,
 * feedback-decision sequence, QPSK map/slice identity and the TX->RX round trips.
 * Values that no reference test fixes (LLR magnitudes, BP iterations, SNR estimates,
 * Schmidl-Cox metric, channel taps, cldpc permutation) are "parity unpinned".
 */
#ifndef DTL_ORACLE_H
#define DTL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_c32;

/* ------------------------------------------------------------------ bits / crc */

/* lib/testbed/repack.cc:31-69 (stateless form: indexes start at 0, as a fresh object) */
int orc_repack_lsb_first(const uint8_t* in, size_t n_in, uint8_t* out,
                         int bits_in, int bits_out, int full_output_symbols);
/* lib/testbed/repack.cc:72-112 */
int orc_repack_msb_first(const uint8_t* in, size_t n_in, uint8_t* out,
                         int bits_in, int bits_out, int full_output_symbols);

/* gr::digital::crc (GR3.10) generic bitwise CRC, as instantiated at
 * lib/dtl/crc_util.cc:23 (32,0x04C11DB7,0xFFFFFFFF,0xFFFFFFFF,refin,refout) and
 * lib/dtl/ofdm_adaptive_packet_header.cc:72 (16,0x1021,0xFFFF,0,false,true). */
uint64_t orc_crc_compute(int num_bits, uint64_t poly, uint64_t init, uint64_t final_xor,
                         int input_reflected, int result_reflected,
                         const uint8_t* data, size_t len);
/* lib/dtl/crc_util.cc:29-36 : CRC-32 appended little-endian */
uint32_t orc_crc32_append(uint8_t* buf, size_t len);
/* lib/dtl/crc_util.cc:41-61 */
int orc_crc32_verify(const uint8_t* buf, size_t len_with_crc);

/* ------------------------------------------------------------------ constellations */

enum { ORC_CNST_UNKNOWN = 0, ORC_CNST_BPSK = 1, ORC_CNST_QPSK = 2, ORC_CNST_PSK8 = 3,
       ORC_CNST_QAM16 = 4 };

/* include/gnuradio/dtl/ofdm_adaptive_utils.h:22-28, lib/dtl/ofdm_adaptive_utils.cc:64-70 */
int orc_bits_per_symbol(int cnst);
/* points: GR3.10 constellation_{bpsk,qpsk,8psk,16qam} + lib/dtl/constellation.cc:18-24 */
int orc_constellation_points(int cnst, orc_c32* pts /* >=16 */);
/* GR3.10 decision_maker of each constellation (used at lib/dtl/ofdm_adaptive_equalizer.cc:253) */
unsigned orc_decision_maker(int cnst, orc_c32 sample);
/* GR3.10 constellation::calc_soft_dec (used at
 * lib/dtl/ofdm_adaptive_constellation_soft_cf_impl.cc:144); out[0..bps) MSB first.
 * dist_mode 0: |x-s|^2 / npwr (SURVEY 8c choice), 1: |x-s| / npwr.
 * ORC_DEMAP_FAST added to dist_mode selects the production arithmetic (16-QAM with squared distances only:
 * the 4 x 4 grid separates, P(bit) needs the 4 levels of ONE axis -- 8 exp instead of 16; other cases unchanged). */
#define ORC_DEMAP_FAST 0x100
void orc_calc_soft_dec(int cnst, orc_c32 sample, float npwr, int dist_mode, float* out);
/* lib/dtl/ofdm_adaptive_constellation_soft_cf_impl.cc:143-148 (+ sigma rule :89-103):
 * n_syms symbols -> n_syms*bps LLRs, LSB first per symbol.  noise<0 => no noise tag. */
void orc_soft_demap(int cnst, const orc_c32* syms, int n_syms, double noise, int dist_mode,
                    float* llr_out);
/* test hooks: the fixed-sequence math of orc_math.h over arrays */
void orc_math_logf(const float* x, long n, float* y);
void orc_math_expf(const float* x, long n, float* y);
void orc_math_sincos_phase(const uint32_t* a, long n, float* sn, float* cs);

/* ------------------------------------------------------------------ LDPC */

typedef struct orc_ldpc orc_ldpc;

/* GR3.10 alist::read + cldpc::set_alist (GF2Mat::get_G) as driven by
 * lib/dtl/ldpc_dec.cc:34-59 / lib/dtl/ldpc_enc.cc:35-56 */
orc_ldpc* orc_ldpc_load(const char* alist_path);
void orc_ldpc_free(orc_ldpc* c);
int orc_ldpc_n(const orc_ldpc* c);
int orc_ldpc_k(const orc_ldpc* c);
int orc_ldpc_m(const orc_ldpc* c);
int orc_ldpc_nedges(const orc_ldpc* c);
void orc_ldpc_permute(const orc_ldpc* c, int* out_n);
/* parity part of G after reduction: rank x k bytes, row-major (for the product's tests) */
void orc_ldpc_parity_matrix(const orc_ldpc* c, uint8_t* out_rank_by_k);
/* lib/dtl/ldpc_enc.cc:58-68 -> GR3.10 cldpc::encode; in: k bits (1/byte), out: n bits */
void orc_ldpc_encode(const orc_ldpc* c, const uint8_t* data_k, uint8_t* out_n);
/* lib/dtl/ldpc_dec.cc:62-71 -> GR3.10 awgn_bp::decode.  llr: n floats (positive = bit 1,
 * on-air order); out: k bits (1/byte).  sigma: awgn_bp's sigma (sqrt(2)
 * makes rx_lr = exp(llr)).  Returns k. */
int orc_ldpc_decode(const orc_ldpc* c, const float* llr_n, int max_it, float sigma,
                    uint8_t* out_k, int* nit);
/* The decoder in PRODUCTION ("fast") arithmetic: the same sum-product recursion with one division per batch of
 * quotients and prefix x suffix extrinsic products (orc_ldpc.c); NOT the reference's bits -- the oracle of the
 * product's `arith = DTL_ARITH_FAST` mode.  arith: 0 = gr_exact (awgn_bp's operations), 1 = fast. */
int orc_ldpc_decode_fast(const orc_ldpc* c, const float* llr_n, int max_it, float sigma,
                         uint8_t* out_k, int* nit);
int orc_ldpc_decode_arith(const orc_ldpc* c, const float* llr_n, int max_it, float sigma, int arith,
                          uint8_t* out_k, int* nit);
void orc_ldpc_decode_batch_arith(const orc_ldpc* c, const float* llr, int ncw, int max_it,
                                 float sigma, int arith, uint8_t* out_bits, int* nit, int n_threads);
/* batch helper (same function, ncw codewords back to back); n_threads>1 uses pthreads */
void orc_ldpc_decode_batch(const orc_ldpc* c, const float* llr, int ncw, int max_it,
                           float sigma, uint8_t* out_bits, int* nit, int n_threads);

/* ------------------------------------------------------------------ packet header */

typedef struct {
    int payload_len;   /* bytes in the frame (12 bits)          */
    int frame_no;      /* 12 bits                               */
    int cnst;          /* 4 bits                                */
    int fb_cnst;       /* 4 bits                                */
    int tb_no;         /* 12 bits  (FEC header only)            */
    int fec_fb;        /* 4 bits                                */
    int tb_offset;     /* 12 bits                               */
    int fec;           /* 4 bits                                */
    int tb_payload;    /* 16 bits                               */
} orc_header;

/* lib/dtl/ofdm_adaptive_packet_header.cc:141-199 (header_formatter, add_fec_header, pack_crc);
 * bits: header_len bytes (1 bit each); header_len = 48 (no FEC) / 96 (FEC).  scramble: the
 * GR3.10 packet_header_ofdm LFSR mask (0x8a, 0x6f, 7) when non-zero. */
void orc_header_format(const orc_header* h, int has_fec, int header_len, int scramble,
                       uint8_t* bits);
/* :231-312 header_parser.  *state_cnst = last valid constellation (starts at QAM16, :70).
 * Returns 1 when the CRC-16 matches (and bps != 0); fills h, *n_syms (len tag, :281-284). */
int orc_header_parse(const uint8_t* bits, int has_fec, int header_len, int scramble,
                     int* state_cnst, orc_header* h, int* n_syms);

/* ------------------------------------------------------------------ feedback decision */

typedef struct {
    int n_mcs;
    double snr_th[16];
    int cnst[16];
    int fec[16];
    double hysteresis;
    int decision_th;
    int decision_counter, last_decision, new_decision;
} orc_feedback;
/* lib/dtl/ofdm_adaptive_feedback_decision.cc:29-52 */
void orc_feedback_init(orc_feedback* f, double hysteresis, int decision_th, int n_mcs,
                       const double* snr_th, const int* cnst, const int* fec, int initial_mcs);
/* :56-96 get_feedback + update_decision */
void orc_feedback_get(orc_feedback* f, double snr, int* cnst, int* fec);

/* ------------------------------------------------------------------ transport blocks (FEC) */

/* lib/dtl/fec_utils.cc:104-112 */
int orc_compute_tb_len(int cw_len, int frame_len);
/* lib/dtl/fec_utils.cc:114-121 */
int orc_align_bits_to_bytes(int nbits);

/* lib/dtl/tb_encoder.cc:32-74: payload bits (1/byte) -> TB bits [ncheck | k_i] per codeword.
 * Returns TB length in bits. */
int orc_tb_encode(const orc_ldpc* c, const uint8_t* bits, int len, int ncws, uint8_t* tb_out);

/* FEC frame builder: restates the steady-state path of
 * lib/dtl/ofdm_adaptive_fec_frame_bvb_impl.cc:284-565 for one large PDU (no pacing, no empty
 * frames, constant MCS): user bytes -> TBs (CRC-32, tb_encode) -> frames of frame_capacity
 * symbols (values 0..2^bps-1) + one orc_header per frame.  Returns number of frames. */
int orc_fec_frames_build(const orc_ldpc* c, int fec_idx, int cnst, int frame_capacity,
                         const uint8_t* user, int n_user, int first_frame_no,
                         uint8_t* syms_out /* max_frames*frame_capacity */,
                         orc_header* hdr_out, int max_frames);

typedef struct orc_tbdec orc_tbdec;
typedef struct {
    int tb_no, tb_payload, crc_ok, avg_it, user_bits;
    long out_offset; /* offset of this TB's user bits in the output stream, -1 if CRC failed */
} orc_tb_rec;
/* ofdm_adaptive_fec_decoder_impl ctor (:34-75) + tb_decoder ctor (tb_decoder.cc:20-29) */
orc_tbdec* orc_tbdec_new(const orc_ldpc* c, int frame_capacity, int max_bps, int max_it,
                         float bp_sigma);
void orc_tbdec_free(orc_tbdec* d);
void orc_tbdec_set_arith(orc_tbdec* d, int arith);
/* test aid: BP iteration count of every codeword the decoder runs (ldpc_dec::decode's *nit, lib/dtl/ldpc_dec.cc:62-71),
 * appended in decode order; *count keeps counting past cap */
void orc_tbdec_set_nit_log(orc_tbdec* d, int* log, long cap, long* count);
/* one frame through ofdm_adaptive_fec_decoder_impl::general_work (:80-217) ->
 * tb_decoder::process_frame (tb_decoder.cc:32-141) -> decode (:143-195) -> on_data_ready.
 * llr: len floats (len = n_syms*bps).  Appends user bits (1/byte) to out (cap out_cap) and TB
 * records to recs.  Returns number of TBs completed by this frame (0..2), <0 on error. */
int orc_tbdec_frame(orc_tbdec* d, const float* llr, int len, int cnst, const orc_header* h,
                    uint8_t* out, long out_cap, long* out_len, orc_tb_rec* recs, int recs_cap,
                    int* n_recs);
/* lib/dtl/ofdm_adaptive_frame_bb_impl.cc:131-163 (no-FEC frame builder, config #1) */
int orc_nofec_frame_build(int cnst, int frame_capacity, const uint8_t* user, int n_user,
                          int frame_no, uint8_t* syms_out, orc_header* h);
/* lib/dtl/ofdm_adaptive_fec_pack_bb_impl.cc:41-62: 8 bits -> byte, LSB first */
long orc_fec_pack(const uint8_t* bits, long nbits, uint8_t* bytes);

/* ------------------------------------------------------------------ OFDM */

typedef struct {
    int fft_len, cp_len, frame_length, header_len, has_fec;
    int n_occ_sets, max_occ;
    const int* occ_len;            /* n_occ_sets                              */
    const int* occ;                /* n_occ_sets x max_occ, signed carrier idx */
    int n_pilot_sets, n_pilots;
    const int* pilot_carriers;     /* n_pilot_sets x n_pilots; ORC_NO_CARRIER pads a set that has fewer pilots (the
                                    * reference's sets may differ in size, qa_ofdm_adaptive_frame_equalizer_vcvc.py:81) */
    int n_pilot_sym_sets;
    const orc_c32* pilot_symbols;  /* n_pilot_sym_sets x n_pilots             */
    const orc_c32* sync1;          /* fft_len, shifted (DC at N/2)            */
    const orc_c32* sync2;
    float sync_threshold;
    float eq_alpha;
    int scramble_header;
    int llr_dist_mode;
    float bp_sigma;
    int max_it;
    int use_sync_correct;
    int arith;                     /* 0: gr_exact (the reference's operations), 1: fast (BP + 16-QAM demapper) */
} orc_cfg;

#define ORC_NO_CARRIER 0x7fffffff
/* python/dtl/ofdm_adaptive.py:1-6 */
int orc_frame_capacity(const orc_cfg* cfg);

/* GR3.10 fft_vcc(N, forward, no window, shift=True) / (N, inverse, (), True): unnormalised */
void orc_fft_shifted(const orc_c32* in, orc_c32* out, int n, int forward);

/* TX: GR3.10 ofdm_carrier_allocator_cvc + fft_vcc(inverse, shift) + ofdm_cyclic_prefixer
 * (python/dtl/ofdm_transmitter.py:168-189).  hdr_bits: header_len*n_occ BPSK bits, payload
 * symbol values (frame_capacity) of constellation cnst.  Writes (2+h+L)*(N+cp) samples. */
void orc_tx_frame(const orc_cfg* cfg, const uint8_t* hdr_bits, const uint8_t* payload_syms,
                  int cnst, orc_c32* out);

/* GR3.10 ofdm_sync_sc_cfb (python/dtl/ofdm_receiver.py:71-72): metric, plateau detector
 * (cp_len, threshold) -> trigger bytes; arg(P) sample-and-held at triggers -> angle. */
void orc_sync_sc(const orc_cfg* cfg, const orc_c32* iq, long n, float* metric /*opt*/,
                 uint8_t* trig, float* angle);

typedef struct {
    int frame_len, remainder;
    long missing_count, correction_count, error_count;
    int acc_error, trigger_counter, synced, in_sync;
} orc_framedetect;
void orc_framedetect_init(orc_framedetect* s, int frame_len);
/* lib/dtl/ofdm_adaptive_frame_detect_bb_impl.cc:64-173 */
void orc_fix_sync(orc_framedetect* s, const uint8_t* in, uint8_t* out, int len);

/* delay(N+cp) + frequency_modulator_fc(-2/N) + multiply_cc (ofdm_receiver.py:73-76,88-89) */
void orc_cfo_mix(const orc_cfg* cfg, const orc_c32* iq, long n, const float* angle, orc_c32* out);
void orc_cfo_mix_trig(const orc_cfg* cfg, const orc_c32* iq, long n, const float* angle, const uint8_t* trig, orc_c32* out);

/* GR3.10 ofdm_chanest_vcvc(sync1, sync2, h) (ofdm_receiver.py:102-103): in = 2 x N (FFT'd sync
 * symbols); out: integer carrier offset + N taps */
int orc_chanest(const orc_cfg* cfg, const orc_c32* sync_syms, orc_c32* taps);

typedef struct {
    double snr_db, noise_db, noise_lin;
} orc_snr;

/* frame equaliser block + equaliser object.
 * payload: lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:99-255 around
 *          ofdm_adaptive_payload_equalizer::equalize (lib/dtl/ofdm_adaptive_equalizer.cc:82-111,
 *          217-268); alpha = cfg->eq_alpha, symbols_skipped = header_len.
 * header : GR3.10 ofdm_frame_equalizer_vcvc around ofdm_adaptive_header_equalizer (:161-174),
 *          with the constructor-argument quirk (:147-154): alpha = 1.0, symbols_skipped = 0.
 * frame: n_sym x N in (FFT output, shifted); hard/soft: n_sym x N out (soft may be NULL);
 * taps_io: N taps in, updated + phase-rotated taps out. */
void orc_frame_equalize(const orc_cfg* cfg, int is_header, int cnst, const orc_c32* frame,
                        int n_sym, int carr_offset, orc_c32* taps_io, orc_c32* hard,
                        orc_c32* soft, orc_snr* snr);

/* GR3.10 ofdm_serializer_vcc (ofdm_receiver.py:122-125,175-180); returns symbols written */
int orc_serialize(const orc_cfg* cfg, const orc_c32* frame, int n_sym, int symbols_skipped,
                  int packet_len, orc_c32* out);

/* ------------------------------------------------------------------ full receive chain */

typedef struct {
    long trigger;          /* trigger index (stream the demux sees)              */
    int header_ok;
    orc_header hdr;
    int n_payload_syms;    /* packet_length tag, QAM symbols                     */
    int carr_offset;
    double snr_db, noise_lin;
    int fb_cnst, fb_fec;   /* feedback decision after this frame                 */
    int lost_frames;       /* cumulative (frame_equalizer_vcvc_impl.cc:124-138)  */
} orc_frame_rec;

typedef struct {
    /* all optional (NULL = not wanted); capacities in frames / elements */
    int max_frames;
    orc_frame_rec* frames;
    uint8_t* trig_raw; uint8_t* trig_fixed; float* angle;   /* n samples each             */
    orc_c32* eq_hard;      /* max_frames x L x N                                            */
    orc_c32* eq_soft;      /* max_frames x L x N                                            */
    orc_c32* soft_syms;    /* max_frames x frame_capacity (serializer output)               */
    float* llr;            /* max_frames x frame_capacity x 4                               */
    uint8_t* hdr_bits;     /* max_frames x header bits                                      */
    orc_c32* taps;         /* max_frames x N  (channel taps handed to the payload equaliser)*/
    uint8_t* user_bits; long user_bits_cap; long n_user_bits;   /* FEC: decoded bits (1/byte) */
    uint8_t* user_bytes; long user_bytes_cap; long n_user_bytes;
    orc_tb_rec* tbs; int tbs_cap; int n_tbs;
    int n_frames;
    int* cw_nit; long cw_nit_cap; long n_cw_nit;   /* iterations per codeword, decode order (optional) */
} orc_rx_out;

/* whole RX graph of python/dtl/ofdm_receiver.py:59-246 over one sample stream */
int orc_rx_stream(const orc_cfg* cfg, const orc_ldpc* code, const orc_feedback* fb_init,
                  const orc_c32* iq, long n, orc_rx_out* out);

#ifdef __cplusplus
}
#endif
#endif
