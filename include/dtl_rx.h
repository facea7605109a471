/*
 * dtl_rx.h -- C ABI of libdtl_hip.so: the MI355X (gfx950) implementation of gr-dtl's
 * adaptive-OFDM receive path.  Plain pointers and sizes only; every "d_" pointer is DEVICE
 * memory owned by the caller; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All functions return 0 (DTL_OK) or a negative dtl_status; dtl_last_error() gives the text.
 * No C++ exception crosses this boundary: where the reference throws out of work() (std::invalid_argument "Missing
 * constellation tag." lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:148-150, std::runtime_error "rcv buffer does not
 * have enough capacity!" lib/dtl/tb_decoder.cc:58-60, "Tags missing" lib/dtl/ofdm_adaptive_fec_decoder_impl.cc:121-126)
 * the entry points return DTL_ERR_TAGS_MISSING / DTL_ERR_CAPACITY; every entry point's body runs inside a handler that
 * turns std::bad_alloc / std::length_error into DTL_ERR_CAPACITY, std::system_error (a helper thread that cannot be
 * started) into DTL_ERR_HIP and anything else into DTL_ERR_INTERNAL; a failed call leaves the handle usable (a receiver's
 * stream state is reset, as after final = 1).
 * There is no CPU fallback: without a HIP device every compute entry point fails with
 * DTL_ERR_NO_DEVICE / DTL_ERR_HIP.
 *
 * Each entry point names the reference interface it replaces (paths relative to gr-dtl).
 */
#ifndef DTL_RX_H
#define DTL_RX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum dtl_status {
    DTL_OK = 0,
    DTL_ERR_INVALID_ARG = -1,
    DTL_ERR_IO = -2,
    DTL_ERR_HIP = -3,
    DTL_ERR_TAGS_MISSING = -4,
    DTL_ERR_CAPACITY = -5,
    DTL_ERR_NO_DEVICE = -6,
    DTL_ERR_UNSUPPORTED = -7,
    DTL_ERR_INTERNAL = -8      /* an unexpected C++ exception was stopped at the boundary (text: dtl_last_error) */
};

/* gr::dtl::constellation_type_t, include/gnuradio/dtl/ofdm_adaptive_utils.h:22-28 */
enum dtl_constellation {
    DTL_CNST_UNKNOWN = 0, DTL_CNST_BPSK = 1, DTL_CNST_QPSK = 2, DTL_CNST_PSK8 = 3, DTL_CNST_QAM16 = 4
};

/* Arithmetic of the belief-propagation decoder and of the 16-QAM soft demapper.
 * GR_EXACT (default): the reference's operations in the reference's order (GR awgn_bp: 4 double divisions per Tanner
 *   edge per iteration; calc_soft_dec: one expf per constellation point) -- hard decisions and iteration counts equal
 *   the reference decoder's.
 * FAST: production arithmetic, algebraically the same sum-product / exact-MAP rules with fewer operations (one division
 *   per batch of a row's quotients, prefix x suffix extrinsic products, bounded messages; separable 16-QAM
 *   likelihoods: 8 expf instead of 16).  Results are bit-identical to the oracle's restatement of THIS arithmetic
 *   (orc_ldpc_decode_fast, ORC_DEMAP_FAST), not to the reference's; BER is no worse. */
enum dtl_arith { DTL_ARITH_GR_EXACT = 0, DTL_ARITH_FAST = 1 };
/* flag added to a dist_mode argument / dtl_rx_cfg.llr_dist_mode: soft demapper in FAST arithmetic */
#define DTL_DEMAP_FAST 0x100

const char* dtl_last_error(void);
const char* dtl_version(void);
/* number of visible HIP devices (0 when none / runtime unusable) */
int dtl_device_count(void);

/* ------------------------------------------------------------------------------------------
 * LDPC decoder  --  replaces gr::dtl::make_ldpc_decoders / ldpc_dec (include/gnuradio/dtl/fec.h:25-37,
 * lib/dtl/ldpc_dec.cc:23-86) and, for the batched form, the per-codeword loop of
 * tb_decoder::decode (lib/dtl/tb_decoder.cc:160-190).
 * ------------------------------------------------------------------------------------------ */
typedef struct dtl_ldpc dtl_ldpc;

/* ldpc_dec::ldpc_dec(alist_fname, max_it): parse the alist, reduce H to systematic form
 * (cldpc), build device tables.  max_iterations: 15 in the reference (ldpc_dec.cc:27); <= 0 selects that default.
 * sigma: awgn_bp's noise sigma, which gr-dtl never sets; pass 0 for the documented default
 * sqrt(2) (rx likelihood ratio = exp(llr)). */
int dtl_ldpc_create(const char* alist_path, int max_iterations, float sigma, dtl_ldpc** out);
int dtl_ldpc_create_from_text(const char* alist_text, size_t text_len, int max_iterations,
                              float sigma, dtl_ldpc** out);
void dtl_ldpc_destroy(dtl_ldpc* h);
/* enum dtl_arith; takes effect from the next decode call */
int dtl_ldpc_set_arith(dtl_ldpc* h, int32_t arith);
int dtl_ldpc_get_n(const dtl_ldpc* h);        /* fec_dec::get_n */
int dtl_ldpc_get_k(const dtl_ldpc* h);        /* fec_dec::get_k */
int dtl_ldpc_get_permute(const dtl_ldpc* h, int32_t* out_n); /* cldpc permutation (print_permute hack) */
/* fec_enc::encode (lib/dtl/ldpc_enc.cc:58-68) on HOST buffers: k bits in (1/byte) -> n bits out,
 * on-air order [n-k check bits | k systematic bits].  Test-vector / TX helper. */
int dtl_ldpc_encode_host(const dtl_ldpc* h, const uint8_t* data_k, uint8_t* out_n);

int dtl_ldpc_encode_batch_host(const dtl_ldpc* h, const uint8_t* data, size_t ncw, uint8_t* out);

/* fec_dec::decode(const float* in, int* nit, unsigned char* out) for ncw codewords:
 * d_llr  : ncw * n floats, positive = bit 1, on-air order (what ldpc_dec::decode receives)
 * d_bits : ncw * k bytes (0/1), the systematic bits
 * d_nit  : ncw int32 iteration counts (0 = initial hard decision already a codeword) */
int dtl_ldpc_decode_batch(const dtl_ldpc* h, const float* d_llr, size_t ncw, uint8_t* d_bits,
                          int32_t* d_nit, void* stream);

/* Gather-table form used by the transport-block path (tb_decoder::decode): codeword i takes
 * len0 soft bits at llr_off0 then len1 at llr_off1; sent positions beyond len0+len1 were never
 * received (soft value 0); positions beyond (n-k)+k_used are shortened (-15); k_used decoded bits
 * are written at out_off. */
typedef struct dtl_cw_desc {
    uint32_t llr_off0, llr_off1, out_off;
    uint16_t len0, len1, k_used, pad;
} dtl_cw_desc;
int dtl_ldpc_decode_gather(const dtl_ldpc* h, const float* d_llr, const dtl_cw_desc* d_desc,
                           size_t ncw, uint8_t* d_bits, int32_t* d_nit, void* stream);

/* ------------------------------------------------------------------------------------------
 * Receiver  --  replaces the RX graph that dtl.ofdm_adaptive_rx / dtl.ofdm_receiver build
 * (python/dtl/ofdm_adaptive_rx.py:32-59, python/dtl/ofdm_receiver.py:11-246): ofdm_sync_sc_cfb,
 * frame_detect_bb, delay/oscillator/mixer, header_payload_demux, fft_vcc, ofdm_chanest_vcvc,
 * header + payload frame equalisers, serializers, soft demapper, FEC decoder, fec_pack
 * (or the hard demapper + frame_pack of the no-FEC tail).  dtl_rx_process = one GR "work" over a
 * self-contained batch of frames; dtl_rx_process_stream continues one sample stream over consecutive calls.
 * ------------------------------------------------------------------------------------------ */
typedef struct dtl_rx dtl_rx;

#define DTL_NO_CARRIER 0x7fffffff
/* dataclass ofdm_adaptive_config / ofdm_adaptive_rx_config (python/dtl/ofdm_adaptive_config.py:10-63) */
typedef struct dtl_rx_cfg {
    int32_t fft_len, cp_len, frame_length, has_fec;
    int32_t n_occ_sets, max_occ;
    const int32_t* occ_len;          /* n_occ_sets */
    const int32_t* occ;              /* n_occ_sets x max_occ, signed carrier indices */
    int32_t n_pilot_sets, n_pilots;
    const int32_t* pilot_carriers;   /* n_pilot_sets x n_pilots; DTL_NO_CARRIER pads a set with fewer pilots */
    int32_t n_pilot_sym_sets;
    const float* pilot_symbols;      /* n_pilot_sym_sets x n_pilots complex (re,im) */
    const float* sync_word1;         /* fft_len complex (re,im), shifted (DC at N/2) */
    const float* sync_word2;
    float sync_threshold, eq_alpha;
    int32_t use_sync_correct, scramble_header;
    int32_t llr_dist_mode;           /* 0: |x-s|^2/npwr (default), 1: |x-s|/npwr (calc_soft_dec variants) */
    int32_t max_iterations;          /* LDPC BP, 15 */
    float bp_sigma;                  /* 0 => sqrt(2) */
    int32_t n_fec_codes;
    const char* const* alist_texts;  /* fec_codes[i] -> FEC id i+1 (ofdm_receiver.py:47) */
    const size_t* alist_lens;
    int32_t n_mcs;                   /* feedback LUT (snr_th, (cnst, fec id)) */
    const double* mcs_snr;
    const int32_t* mcs_cnst;
    const int32_t* mcs_fec;
    int32_t initial_mcs;
    int32_t decision_th;             /* 5 (ofdm_receiver.py:167) */
    double hysteresis;               /* 1 */
    int32_t arith;                   /* enum dtl_arith: decoder + 16-QAM demapper arithmetic (0 = the reference's) */
    int32_t device;                  /* HIP device the handle lives on: its tables, buffers and every kernel launch.  Each entry
                                      * point makes it current for the duration of the call (and restores the caller's device);
                                      * the caller's device pointers and stream must belong to it.  DTL_DEVICE_CURRENT: whatever
                                      * device is current in the calling thread (one process per GPU with the device set once) */
} dtl_rx_cfg;
#define DTL_DEVICE_CURRENT (-1)

/* header fields = the "header_data" dict of packet_headerparser_b; snr/noise/feedback = what
 * ofdm_adaptive_frame_equalizer_vcvc publishes per frame (tags, feedback_port, monitor) */
typedef struct dtl_rx_frame_info {
    int64_t trigger;
    int32_t header_ok, carr_offset;
    int32_t payload_len, frame_no, cnst, fb_cnst, tb_no, fec_fb, tb_offset, fec, tb_payload;
    int32_t n_payload_syms, payload_index;
    double snr_db, noise;
    int32_t feedback_cnst, feedback_fec, lost_frames, cand;
    int64_t llr_offset;
} dtl_rx_frame_info;

/* monitor message of ofdm_adaptive_fec_decoder (fec_decoder_impl.cc:183-195), per transport block */
typedef struct dtl_rx_tb_info {
    int32_t tb_no, tb_payload, crc_ok, avg_it, user_bits, pad;
    int64_t out_offset;
} dtl_rx_tb_info;

enum dtl_rx_flags {
    DTL_RX_WANT_EQ_HARD = 1,    /* out3 of ofdm_receiver: equalised + sliced symbols, L x N per frame */
    DTL_RX_WANT_EQ_SOFT = 2,    /* out4 before the serializer: pre-slice symbols, L x N per frame     */
    DTL_RX_WANT_SOFT_SYMS = 4,  /* serializer output (frame_capacity per frame)                        */
    DTL_RX_WANT_HDR_BITS = 8,
    DTL_RX_WANT_METRIC = 16,    /* Schmidl-Cox metric per sample (debug)                               */
    DTL_RX_WANT_KERNEL_MS = 32, /* stage_ms[6], [7] (device time of the LDPC / phase-B kernels) read inside the call */
    DTL_RX_WANT_KERNEL_SPLIT = 64 /* HIP events around EVERY kernel launch of the call; read with dtl_rx_kernel_split.  A
                                 * measurement aid: the extra events open small gaps between kernels, not for timed steps */
};

/* All pointers are owned by the handle and stay valid until the next dtl_rx_process / destroy.
 * Host arrays: frames, tbs, raw_triggers, fixed_triggers, trigger_angles.  d_*: device. */
typedef struct dtl_rx_result {
    int64_t n_frames, n_payload_frames, n_tbs, n_payload_bytes, n_raw_triggers, n_fixed_triggers;
    int64_t n_llr, n_codewords;
    const dtl_rx_frame_info* frames;
    const dtl_rx_tb_info* tbs;
    const int64_t* raw_triggers;
    const int64_t* fixed_triggers;
    const float* trigger_angles;
    const uint8_t* d_payload;   /* out0: packed user bytes of the CRC-ok TBs (or of every frame, no FEC) */
    const float* d_eq_hard;     /* complex (re,im), n_payload_frames x L x N */
    const float* d_eq_soft;
    const float* d_soft_syms;   /* n_payload_frames x frame_capacity */
    const float* d_llr;         /* soft bits, frame i at frames[..].llr_offset */
    const float* d_taps;        /* per visited trigger: N taps handed to the payload equaliser */
    const uint8_t* d_hdr_bits;
    const float* d_metric;
    const uint8_t* d_cw_bits;   /* decoded bits of the transport blocks, packed LSB first (1 byte per bit for codes the
                                 * node-parallel kernel does not serve) */
    const int32_t* d_cw_nit;
    double stage_ms[8];
    int32_t frame_capacity, hdr_bits, frame_samples, pad;
} dtl_rx_result;

/* Threading: a dtl_ldpc handle may decode from several host threads / HIP streams at once (every launch has its own
 * work-queue counter; dtl_ldpc_set_arith must not run concurrently with a decode).
 * dtl_rx / dtl_tx: one handle serves one call at a time (it owns its device buffers, staging memory and stream
 * state); different handles may run concurrently from different host threads on different HIP streams
 * (two receivers decoding alternate batches overlap one batch's host phases with the other's decoder kernels). */
int dtl_rx_create(const dtl_rx_cfg* cfg, dtl_rx** out);
void dtl_rx_destroy(dtl_rx* h);
/* d_iq: n_samples complex64 (re,im) on the device */
int dtl_rx_process(dtl_rx* h, const float* d_iq, int64_t n_samples, uint32_t flags, void* stream,
                   dtl_rx_result* out);
/* Streaming form -- what the reference hier block's work() calls amount to (python/dtl/ofdm_receiver.py:59-246
 * run by the GNU Radio scheduler on consecutive buffers): each call continues the sample stream of the
 * previous one.  The Schmidl-Cox/plateau detector, the fine-CFO oscillator, frame_detect_bb, the demux
 * position, the header parser's "previous MCS", the transport block being received (tb_decoder.cc) and
 * the feedback decision keep their state; samples of frames that are not complete yet are kept on the
 * device and re-joined with the next call's samples.  Trigger positions in the result are absolute
 * (samples since the last reset).  final != 0 ends the stream the way dtl_rx_process ends its batch
 * (the detector closes an open plateau, incomplete frames are dropped) and resets the state;
 * dtl_rx_process(h, x, n) == dtl_rx_reset(h) + dtl_rx_process_stream(h, x, n, final = 1).
 * Any number of calls that cut the same stream at arbitrary points returns the same frames, transport
 * blocks and payload bytes as one batch call, except that frame_detect_bb -- as in the reference --
 * looks ahead only to the end of the buffer it is given (use_sync_correct = 0 removes that caveat). */
int dtl_rx_process_stream(dtl_rx* h, const float* d_iq, int64_t n_samples, uint32_t flags, int32_t final,
                          void* stream, dtl_rx_result* out);
/* dtl_rx_process in two halves, for callers that decode batch after batch: dtl_rx_submit hands a self-contained batch
 * over and returns at once (at most two may be waiting), dtl_rx_wait returns the oldest one's result -- what
 * dtl_rx_process would have returned for it, bit for bit.
 *     submit(b0); submit(b1); wait -> r0; submit(b2); wait -> r1; ...
 * The two batches in flight run side by side on two LANES inside the handle (each a receiver with the handle's
 * configuration, a host thread and a stream of its own; the second is made at the first submit that needs it and doubles the
 * handle's device memory): where one's step leaves the device idle -- the host round trips at its front, launch gaps, the
 * decoder's tail -- the other's kernels run.  In the reference's terms this is what GNU Radio's scheduler does by
 * running every block in its own thread: ofdm_sync_sc_cfb works on the next buffer while the FEC decoder works on the
 * previous one (python/dtl/ofdm_receiver.py:71-98 vs :196-211).
 * `stream`: the lane's work starts behind whatever the caller has queued there at submit; d_iq must stay valid and
 * unchanged until the dtl_rx_wait that returns the batch.  A result (host arrays and device buffers) stays valid until the
 * SECOND dtl_rx_submit after its dtl_rx_wait (the lane that produced it is reused then).  DTL_ERR_CAPACITY: two batches
 * are waiting already.  Not to be mixed with a running stream (dtl_rx_process_stream without final) -- plain calls are
 * refused while batches wait -- and every batch starts from a reset, like dtl_rx_process. */
int dtl_rx_submit(dtl_rx* h, const float* d_iq, int64_t n_samples, uint32_t flags, void* stream);
int dtl_rx_wait(dtl_rx* h, dtl_rx_result* out);
void dtl_rx_reset(dtl_rx* h);
/* samples consumed since the last reset */
int64_t dtl_rx_stream_position(const dtl_rx* h);
/* Device time (ms) of the LDPC kernel(s) and of the phase-B kernels (frame equaliser + soft demapper) of the call `back`
 * calls ago (0 = the last one; the handle keeps HIP events of its last 32 calls, recorded on each call's stream);
 * -1 where that call launched none.  Reading after a series of calls keeps the event queries out of the calls. */
int dtl_rx_kernel_ms(dtl_rx* h, int32_t back, double* ldpc_ms, double* payload_ms);
/* Device time per kernel of the last call made with DTL_RX_WANT_KERNEL_SPLIT, summed over the kernel's launches in that call
 * (HIP events recorded on the call's stream around each launch): which kernel is the longest is measured, not assumed. */
enum dtl_rx_kernel_id {
    DTL_K_SYNC_EDGES = 0, DTL_K_HEADER = 1, DTL_K_FRAME_EQ = 2, DTL_K_DEMAP = 3, DTL_K_LDPC = 4, DTL_K_TB_CHECK = 5, DTL_K_COUNT = 6
};
typedef struct dtl_rx_kernel_times {
    double ms[DTL_K_COUNT];         /* sum over the launches; DEMAP: the soft demapper, or the no-FEC tail kernel */
    int32_t launches[DTL_K_COUNT];
    double longest_launch_ms[DTL_K_COUNT];
} dtl_rx_kernel_times;
int dtl_rx_kernel_split(dtl_rx* h, dtl_rx_kernel_times* out);

/* ---------------------------------------------------------------------------------------------------
 * Batched transmitter (the frame generator).  Replaces the TX hier block of
 * python/dtl/ofdm_transmitter.py:63-189 for one PDU and a constant MCS: ofdm_adaptive_fec_frame_bvb
 * (lib/dtl/ofdm_adaptive_fec_frame_bvb_impl.cc:284-565) + tb_encoder (lib/dtl/tb_encoder.cc:32-74) + ldpc_enc
 * (lib/dtl/ldpc_enc.cc:58-68), or ofdm_adaptive_frame_bb (lib/dtl/ofdm_adaptive_frame_bb_impl.cc:131-163);
 * header formatter (lib/dtl/ofdm_adaptive_packet_header.cc:141-199); chunks_to_symbols, carrier allocator,
 * inverse FFT, cyclic prefix.  Uses the carrier / pilot / sync-word / FEC fields of dtl_rx_cfg.
 * ------------------------------------------------------------------------------------------------- */
typedef struct dtl_tx dtl_tx;

typedef struct dtl_tx_frame_info {   /* the packet-header fields of one transmitted frame */
    int32_t payload_len, frame_no, cnst, fb_cnst, tb_no, fec_fb, tb_offset, fec, tb_payload, pad;
} dtl_tx_frame_info;

/* Pointers are owned by the handle and stay valid until the next dtl_tx_process / destroy. */
typedef struct dtl_tx_result {
    int64_t n_frames, n_samples, n_tbs;
    const dtl_tx_frame_info* frames;
    const float* d_iq;            /* n_samples complex64 on the device: frames back to back (+ gap zeros each) */
    const uint8_t* d_syms;        /* optional: n_frames x frame_capacity constellation indices */
    const uint8_t* d_hdr_bits;    /* n_frames x hdr_bits header bits (1 bit per byte) */
    int32_t frame_capacity, hdr_bits, frame_samples, pad;
} dtl_tx_result;

int dtl_tx_create(const dtl_rx_cfg* cfg, dtl_tx** out);
void dtl_tx_destroy(dtl_tx* h);
/* d_user: n_user bytes on the device; constellation 1..4; fec = FEC id (0 / ignored without FEC);
 * first_frame_no / first_tb_no continue the numbering of a previous call (mixed-MCS streams);
 * gap = zero samples appended to every frame; want_syms != 0 also returns the symbol indices. */
/* The constellation tables the kernels use (host only): gr::digital constellation_{bpsk,qpsk,8psk,16qam} with
 * gr-dtl's 0.5 scaling of QPSK (lib/dtl/constellation.cc:18-24, lib/dtl/ofdm_adaptive_utils.cc:52-70); 16 complex
 * (re, im) slots, *n_points of them valid */
int dtl_constellation_points_host(int32_t constellation, float* points16, int32_t* n_points);
/* ofdm_adaptive_packet_header::header_formatter (lib/dtl/ofdm_adaptive_packet_header.cc:141-199) on the host: the
 * header bits (1 per byte, header_len_bits of them) the transmitter puts on the header symbols */
int dtl_header_format_host(const dtl_tx_frame_info* fields, int32_t has_fec, int32_t header_len_bits, uint8_t* bits_out);
/* The framing state machine alone (host only, no device needed): the packet-header fields of the frames that
 * dtl_tx_process would emit for n_user bytes.  *n_out = frames; DTL_ERR_CAPACITY when cap is too small. */
int dtl_tx_plan(dtl_tx* h, int64_t n_user, int32_t constellation, int32_t fec, int32_t first_frame_no,
                int32_t first_tb_no, dtl_tx_frame_info* frames_out, int64_t cap, int64_t* n_out);
int dtl_tx_process(dtl_tx* h, const uint8_t* d_user, int64_t n_user, int32_t constellation, int32_t fec,
                   int32_t first_frame_no, int32_t first_tb_no, int32_t gap, int32_t want_syms, void* stream,
                   dtl_tx_result* out);

/* ---------------------------------------------------------------------------------------------------
 * Single stages of the receive graph, for block shims and unit tests (the receiver above runs fused kernels
 * with the same device arithmetic).  Device pointers unless named *_host.
 * ------------------------------------------------------------------------------------------------- */
/* GR ofdm_sync_sc_cfb as wired at python/dtl/ofdm_receiver.py:71-72 + sample_and_hold of arg(P): optional
 * timing metric per sample (d_metric, n floats, may be NULL), trigger positions and the angle held at each
 * trigger (host arrays of capacity `cap`; *n_out = triggers found, DTL_ERR_CAPACITY when cap is too small) */
int dtl_sync_sc(const float* d_iq, int64_t n_samples, int32_t fft_len, int32_t cp_len, float threshold,
                float* d_metric, int64_t* triggers_host, float* angles_host, int64_t cap, int64_t* n_out,
                void* stream);
/* GR fft_vcc(fft_len, forward, (), shift = true) on n_vectors vectors of fft_len complex samples
 * (python/dtl/ofdm_receiver.py:116-121, python/dtl/ofdm_transmitter.py:172-178); unnormalised */
int dtl_fft_batch(const float* d_in, float* d_out, int64_t n_vectors, int32_t fft_len, int32_t forward, void* stream);
/* ofdm_adaptive_constellation_soft_cf (lib/dtl/ofdm_adaptive_constellation_soft_cf_impl.cc:89-148): n symbols of
 * one constellation, noise = the equaliser's linear noise estimate (< 0: none, sigma 0.001) -> bits/symbol soft
 * bits per symbol, LSB first; dist_mode as in dtl_rx_cfg.llr_dist_mode */
int dtl_soft_demap(const float* d_syms, int64_t n_syms, int32_t constellation, double noise, int32_t dist_mode,
                   float* d_llr, void* stream);
/* CRC-32 of a device byte string, as lib/dtl/crc_util.cc:41-61 computes it (crc(32, 0x04C11DB7, ~0, ~0, refin,
 * refout)); the value is returned to the host (the call synchronises `stream`) */
int dtl_crc32(const uint8_t* d_bytes, int64_t len, uint32_t* crc_host, void* stream);

/* ---- the blocks between the FFT and the soft demapper, on the frequency-domain vectors the reference's blocks
 * exchange (fft_vcc output: fft_len complex per OFDM symbol, shifted, DC at fft_len/2).  The handle supplies the
 * configuration (carrier maps, pilots, sync words, equaliser alpha ...); the calls are stateless. */
/* digital.ofdm_chanest_vcvc(sync_word1, sync_word2, header_len) as wired at python/dtl/ofdm_receiver.py:102-103.
 * d_sync_syms: n x 2 x fft_len (the two synchronisation symbols of n frames) -> per frame the integer carrier
 * offset (tag "ofdm_sync_carr_offset") and fft_len initial channel taps (tag "ofdm_sync_chan_taps"), on the device */
int dtl_chanest(dtl_rx* h, const float* d_sync_syms, int64_t n, int32_t* d_carr_offset, float* d_taps, void* stream);

/* raw header fields of one frame + CRC-16 flag (the "header_data" message of packet_headerparser_b) */
typedef struct dtl_header_fields {
    int32_t crc_ok, carr_offset;
    int32_t payload_len, frame_no, cnst, fb_cnst, tb_no, fec_fb, tb_offset, fec, tb_payload, pad;
} dtl_header_fields;
/* The header path behind the channel estimator (python/dtl/ofdm_receiver.py:104-150): ofdm_frame_equalizer_vcvc around
 * gr::dtl::ofdm_adaptive_header_equalizer (lib/dtl/ofdm_adaptive_equalizer.cc:161-174), ofdm_serializer_vcc, BPSK
 * constellation_decoder_cb, packet_headerparser_b with ofdm_adaptive_packet_header::header_parser
 * (lib/dtl/ofdm_adaptive_packet_header.cc:231-312).  d_hdr_syms: n x header_len x fft_len; d_taps / d_carr_offset as
 * dtl_chanest returns them.  fields_host[n]: the parsed fields as transmitted (the parser's "keep the previous
 * constellation when the field is invalid" rule, :269-273, is sequential state and stays with the caller).
 * Optional (may be NULL): d_hdr_bits n x header bits (1 per byte); d_taps_out n x fft_len taps handed on to the
 * payload equaliser.  Synchronises `stream`. */
int dtl_header_decode(dtl_rx* h, const float* d_hdr_syms, const float* d_taps, const int32_t* d_carr_offset, int64_t n,
                      dtl_header_fields* fields_host, uint8_t* d_hdr_bits, float* d_taps_out, void* stream);

/* what the frame equaliser block reads from tags (constellation, "ofdm_sync_carr_offset", packet length) */
typedef struct dtl_frame_eq_in {
    int32_t cnst, carr_offset, n_payload_syms, pad;
} dtl_frame_eq_in;
/* what it publishes per frame (estimated_snr_tag_key / noise_tag_key, lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:195-216) */
typedef struct dtl_frame_eq_out {
    double snr_db, noise_db, noise;
} dtl_frame_eq_out;
/* gr::dtl::ofdm_adaptive_frame_equalizer_vcvc::make(equalizer, feedback_decision, cp_len, ...) -> work
 * (include/gnuradio/dtl/ofdm_adaptive_frame_equalizer_vcvc.h:30-36, lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:99-255)
 * around ofdm_adaptive_payload_equalizer::equalize (lib/dtl/ofdm_adaptive_equalizer.cc:217-268), plus the
 * ofdm_serializer_vcc behind it (python/dtl/ofdm_receiver.py:175-180), for n frames:
 * d_frames: n x frame_length x fft_len; d_taps: n x fft_len (from dtl_header_decode); in_host[n].
 * Outputs (each may be NULL): d_eq_hard / d_eq_soft n x frame_length x fft_len (out0 / out1 of the block: sliced and
 * pre-slice symbols), d_soft_syms n x frame_capacity (serializer output, zero behind the frame's last symbol),
 * out_host[n].  DTL_ERR_TAGS_MISSING ("Missing constellation tag.", :148-150) for a constellation outside 1..4.
 * Synchronises `stream`. */
int dtl_equalize_frames(dtl_rx* h, const float* d_frames, const float* d_taps, const dtl_frame_eq_in* in_host, int64_t n,
                        float* d_eq_hard, float* d_eq_soft, float* d_soft_syms, dtl_frame_eq_out* out_host, void* stream);

/* one completed transport block of dtl_tb_gather: codewords cw_first .. cw_first + n_cw - 1 of the descriptor table,
 * decoded bits (1 per byte) at bits_off, tb_payload_bits of them (user bits + CRC-32) */
typedef struct dtl_tb_span {
    int32_t tb_no, fec, tb_payload_bits, n_cw;
    uint32_t cw_first, bits_off;
} dtl_tb_span;
/* gr::dtl::ofdm_adaptive_fec_decoder::make(decoders, frame_capacity, max_bps, len_key) -> work
 * (include/gnuradio/dtl/ofdm_adaptive_fec_decoder.h:27, lib/dtl/ofdm_adaptive_fec_decoder_impl.cc:80-217) ->
 * tb_decoder::process_frame (lib/dtl/tb_decoder.cc:32-141) and the codeword split of tb_decoder::decode (:143-190),
 * as index arithmetic: which soft bits of the stream every codeword takes.  frames_host: the frames of one
 * self-contained run, in order (header fields, cnst, n_payload_syms, llr_offset = position of the frame's first soft
 * bit in the soft-bit stream; frames with header_ok == 0 or llr_offset < 0 are skipped).  code_n / code_k: n_codes
 * entries, FEC id i+1 -> entry i.  Writes one dtl_cw_desc per codeword to d_cw_desc (device, capacity cw_cap) for
 * dtl_ldpc_decode_gather (codewords of different codes are NOT regrouped: use spans[].fec) and one dtl_tb_span per
 * completed block to spans_host (d_cw_desc == NULL: the block records only, no device work).  DTL_ERR_CAPACITY when a capacity is too small ("rcv buffer does not have enough
 * capacity!" likewise), DTL_ERR_TAGS_MISSING for a frame without a valid constellation, DTL_ERR_UNSUPPORTED for a block
 * spread over more pieces of the stream than a descriptor pair holds (frames whose bit capacity is no whole number of
 * bytes leave a gap behind every frame; dtl_rx_process copies such blocks into one run first, this entry point does not
 * own the soft-bit stream and cannot).  Synchronises `stream`. */
int dtl_tb_gather(const int32_t* code_n, const int32_t* code_k, int32_t n_codes, int32_t frame_capacity, int32_t max_bps,
                  const dtl_rx_frame_info* frames_host, int64_t n_frames, dtl_cw_desc* d_cw_desc, int64_t cw_cap,
                  int64_t* n_cw, dtl_tb_span* spans_host, int64_t span_cap, int64_t* n_spans, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Frame-batch sharding over the GPUs of one node (host arithmetic only: no device, no communication library).
 * Frames are independent given their own preamble and header -- the reference resets the equaliser's taps and its SNR
 * estimator for every frame (lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:117-118,180) -- so one stream of
 * back-to-back frames is cut into contiguous runs, one per GPU, each decoded by its own receiver handle
 * (dtl_rx_cfg.device) with a few frames of context on either side; the caller in the reference is the hier block
 * python/dtl/ofdm_adaptive_rx.py:42-59, which a multi-GPU host instantiates once per device.
 * ------------------------------------------------------------------------------------------------- */
/* samples [start, stop) are what the rank decodes; a transport block belongs to the rank whose [own_lo, own_hi) holds
 * the trigger of the frame it starts in */
typedef struct dtl_shard {
    int64_t start, stop, own_lo, own_hi;
} dtl_shard;
/* n_frames frames of frame_samples samples follow `lead` samples of lead-in in a stream of total_samples; rank r of
 * world_size owns frames [floor(F r / G), floor(F (r + 1) / G)).  Decoded range: max(guard_frames, 4) frames in front
 * (frame_detect_bb must be in sync -- three consistent triggers -- before the rank's own frames begin), guard_frames
 * behind (a transport block that starts in the rank's last frame must be complete: 3 for blocks of about one frame). */
int dtl_shard_plan(int64_t n_frames, int64_t frame_samples, int64_t lead, int64_t total_samples, int32_t world_size,
                   int32_t rank, int32_t guard_frames, dtl_shard* out);
/* Cut a shard into at most n_pieces consecutive pieces for a pipelined hand-over (receive piece k + 1 while
 * dtl_rx_process_stream decodes piece k: the streaming call returns the same frames whatever the cuts).  bounds:
 * n_pieces + 1 entries; piece j is [bounds[j], bounds[j + 1]), *n_out pieces; cuts lie at multiples of `align` samples
 * behind the shard's start (4096 = the sync kernel's tile). */
int dtl_shard_pieces(const dtl_shard* plan, int32_t n_pieces, int64_t align, int64_t* bounds, int32_t* n_out);
/* Indices into `tbs` of the transport blocks this rank owns.  frames / tbs: the records of the rank's decode (one
 * dtl_rx_process call, or the concatenation over the dtl_rx_process_stream calls of its pieces), trigger positions
 * relative to plan->start.  *n_out = count; DTL_ERR_CAPACITY when cap is too small. */
int dtl_owned_blocks(const dtl_rx_frame_info* frames, int64_t n_frames, const dtl_rx_tb_info* tbs, int64_t n_tbs,
                     const dtl_shard* plan, int64_t* idx_out, int64_t cap, int64_t* n_out);
/* Byte ranges (offset, length) of the decoded output that hold the CRC-ok blocks among idx[0..n_idx), adjacent blocks
 * merged -- usually ONE range: what a rank sends to the root. */
int dtl_owned_payload_ranges(const dtl_rx_tb_info* tbs, int64_t n_tbs, const int64_t* idx, int64_t n_idx, int64_t* offs,
                             int64_t* lens, int64_t cap, int64_t* n_out);

/* device-to-device copy on `stream` (lets a host binding snapshot result buffers it does not own) */
int dtl_copy_device(void* d_dst, const void* d_src, size_t bytes, void* stream);

/* Sequential integer helpers exported for unit tests (bit-exact targets):
 * ofdm_adaptive_frame_detect_bb_impl::fix_sync on a sparse trigger list */
int dtl_fix_sync(int32_t frame_len, const int64_t* triggers, int64_t n_triggers, int64_t stream_len,
                 int64_t* out, int64_t out_cap, int64_t* n_out);
/* Test aid for the sync kernel's single-precision pre-test (rx_kernels.hip, pass C): every strip of 8 (16) outputs is
 * run through BOTH the pre-test and the exact double-precision test.  counts_host[0] = strips, [1] = strips the
 * pre-test left to the exact path, [2] = outputs the pre-test decided differently from the exact test (must be 0). */
int dtl_sync_pretest_check(const float* d_iq, int64_t n, int32_t fft_len, float threshold, uint64_t* counts_host, void* stream);
/* the same over consecutive work() buffers (buffer b holds items [buf_ends[b-1], buf_ends[b]) of one trigger stream;
 * the block's state carries over), with every buffer's triggers handed to the walk piecewise: `horizons` (ascending
 * absolute positions) are the points up to which the trigger list is declared complete before the buffer's length is
 * known -- the resumable form the receive path runs under its sync kernels.  Must equal the one-shot form. */
int dtl_fix_sync_chunked(int32_t frame_len, const int64_t* triggers, int64_t n_triggers, const int64_t* buf_ends,
                         int64_t n_bufs, const int64_t* horizons, int64_t n_horizons, int64_t* out, int64_t out_cap,
                         int64_t* n_out);
/* GR plateau_detector_fb(max_len, threshold) over a HOST metric array (the same host routine
 * the pipeline runs on the kernel's threshold crossings) */
int dtl_plateau_detect(const float* metric_host, int64_t n, float threshold, int32_t max_len,
                       int64_t* out, int64_t out_cap, int64_t* n_out);
/* ofdm_adaptive_feedback_decision::get_feedback over a sequence of SNR estimates */
int dtl_feedback_sequence(int32_t n_mcs, const double* mcs_snr, const int32_t* mcs_cnst, const int32_t* mcs_fec,
                          double hysteresis, int32_t decision_th, int32_t initial_mcs, const double* snr,
                          int64_t n, int32_t* out_cnst, int32_t* out_fec);

#ifdef __cplusplus
}
#endif
#endif /* DTL_RX_H */
