// rx_pipeline.h -- host side of the batched receive pipeline: configuration, device tables, the
// sequential control logic the reference keeps in single-threaded blocks (plateau detector,
// frame_detect_bb, header/payload demux, header-parser state, transport-block reassembly,
// feedback decision) and the kernel orchestration.
#pragma once

#include "dtl_common.h"
#include "ldpc_bp.h"
#include "rx_types.h"

#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace dtl {

struct McsEntry {
    double snr_th;
    int cnst;
    int fec;
};

struct RxConfig {
    int fft_len = 64, cp_len = 16, frame_length = 20;
    bool has_fec = false;
    std::vector<std::vector<int>> occupied_carriers;
    std::vector<std::vector<int>> pilot_carriers;
    std::vector<std::vector<cf32>> pilot_symbols;
    std::vector<cf32> sync_word1, sync_word2;
    float sync_threshold = 0.95f;
    float eq_alpha = 0.1f;
    bool use_sync_correct = true;
    bool scramble_header = false;
    int llr_dist_mode = 0;
    int max_iterations = 15;
    float bp_sigma = 0.0f; // 0 => sqrt(2)
    int arith = 0;         // DTL_ARITH_GR_EXACT / DTL_ARITH_FAST: decoder + 16-QAM demapper arithmetic
    int device = -1;       // HIP device of the handle (dtl_rx_cfg.device); -1: the calling thread's current device
    std::vector<std::string> alist_texts; // fec_codes, index i -> FEC id i+1
    std::vector<McsEntry> mcs;
    int initial_mcs = 0;
    double hysteresis = 1.0;  // ofdm_receiver.py:167 ofdm_adaptive_feedback_decision(1, 5, ...)
    int decision_th = 5;
    int header_len() const { return has_fec ? 2 : 1; } // ofdm_receiver.py:66-68
    // Bits of one packet header: the reference builds its formatter from occupied_carriers[0] alone, once per header symbol
    // (python/dtl/ofdm_receiver.py:129-130, python/dtl/ofdm_transmitter.py:76-78)
    int header_bits() const { return header_len() * int(occupied_carriers.empty() ? 0 : occupied_carriers[0].size()); }
    // -> DTL_OK, or DTL_ERR_INVALID_ARG when the header symbols' own carrier sets (symbol i uses set i mod n_sets in the
    // serializer / carrier allocator) do not hold exactly header_bits(): packet_headerparser_b consumes header_bits() items
    // per header while ofdm_serializer_vcc delivers the sets' sum per frame -- parser and demux would fall out of step in
    // the reference; here such a plan is refused when the handle is created
    int check_header_plan() const;
};

// makes the handle's device current for the duration of a call and restores the caller's afterwards
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    int enter(int device);   // device < 0: nothing to do
    ~DeviceScope();
};

// lib/dtl/ofdm_adaptive_frame_detect_bb_impl.cc state (d_* members)
struct FrameDetectState {
    int frame_len = 0, remainder = 0;
    long missing_count = 0, correction_count = 0, error_count = 0;
    int acc_error = 0, trigger_counter = 0, synced = 0;
    bool in_sync = false;
};
// sparse-trigger form of ofdm_adaptive_frame_detect_bb_impl::fix_sync (one buffer of `len` items)
std::vector<int64_t> fix_sync_sparse(FrameDetectState& s, const std::vector<int64_t>& triggers, int64_t len);
// the same walk, resumable: triggers are fed as they become known (ascending; complete below `horizon`), the buffer's
// length is given last.  feed()s + finish(len) == fix_sync_sparse(all triggers, len), state included.
class FixSyncRun {
public:
    explicit FixSyncRun(FrameDetectState& st);
    void feed(const int64_t* trig, size_t n, int64_t horizon);
    std::vector<int64_t> finish(int64_t len);
    int64_t position() const { return i; }   // the walk has dealt with everything in front of this item
private:
    static constexpr int64_t kInf = INT64_MAX;
    void advance(bool final, int64_t len);
    void reseat(int64_t pos);
    int64_t next_set(int64_t pos);
    void insert_added(int64_t pos);
    void erase_at(int64_t pos);
    FrameDetectState& s;
    std::vector<int64_t> triggers, added;    // added: kept sorted
    std::vector<uint8_t> erased;
    size_t cur = 0, cur_add = 0;
    int64_t i = 0, index = 0, last_trigger_index = 0, horizon = 0;
    bool trigger_found = false;
};
// plateau_detector_fb(max_len, threshold) from the metric's threshold crossings
std::vector<int64_t> plateau_from_edges(std::vector<unsigned long long> edges, int64_t n, int max_len);

// lib/dtl/ofdm_adaptive_feedback_decision.cc
struct FeedbackDecision {
    std::vector<McsEntry> lut;
    double hysteresis = 1;
    int decision_th = 5, counter = 0, last = 0, pending = 0;
    void init(const std::vector<McsEntry>& mcs, double hyst, int th, int initial);
    void get(double snr, int* cnst, int* fec);
};

struct FrameInfo { // mirrors dtl_rx_frame_info of the C ABI
    int64_t trigger;
    int32_t header_ok, carr_offset;
    int32_t payload_len, frame_no, cnst, fb_cnst, tb_no, fec_fb, tb_offset, fec, tb_payload;
    int32_t n_payload_syms, payload_index;
    double snr_db, noise;
    int32_t feedback_cnst, feedback_fec, lost_frames, cand;
    int64_t llr_offset;
};

struct TbInfo { // mirrors dtl_rx_tb_info
    int32_t tb_no, tb_payload, crc_ok, avg_it, user_bits, pad;
    int64_t out_offset; // byte offset in the payload output, -1 when the CRC failed
};

enum ProcessFlags : unsigned {
    RX_WANT_EQ_HARD = 1u, RX_WANT_EQ_SOFT = 2u, RX_WANT_SOFT_SYMS = 4u, RX_WANT_HDR_BITS = 8u,
    RX_WANT_METRIC = 16u, RX_WANT_KERNEL_MS = 32u, RX_WANT_KERNEL_SPLIT = 64u,
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~DevBuf();
    int ensure(size_t bytes);
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// page-locked host staging (async copies from/to pageable memory make the runtime pin pages per copy)
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool device_reads = false;  // set before ensure(): kernels read the buffer in place (coherent mapping)
    bool mapped_coherent = false;
    ~HostBuf();
    int ensure(size_t bytes);
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// one completed transport block of tb_gather_frames (mirrors dtl_tb_span of the C ABI)
struct TbSpan {
    int32_t tb_no, fec, tb_payload_bits, n_cw;
    uint32_t cw_first, bits_off;
};
int tb_gather_frames(const std::vector<std::pair<int, int>>& code_nk, int capacity, int max_bps, const FrameInfo* frames,
                     int64_t n_frames, std::vector<TbGather>& gathers, std::vector<TbSpan>& spans);

// Per-kernel device times of ONE call (RX_WANT_KERNEL_SPLIT): an event in front of and behind every launch, on the
// launch stream, from a pool kept by the handle; read after the call's final synchronisation.
struct KernelClock {
    enum Kind { SYNC = 0, HEADER, FRAME_EQ, DEMAP, LDPC, TB_CHECK, N_KINDS };
    bool on = false;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    struct Span { int kind; size_t a, b; };
    std::vector<Span> spans;
    double ms[N_KINDS] = { 0 }, longest[N_KINDS] = { 0 };
    int launches[N_KINDS] = { 0 };
    bool valid = false;
    ~KernelClock();
    void begin(bool enable);                       // start of a call
    int mark(hipStream_t st, size_t* idx);         // record the next event of the pool
    void span(int kind, size_t a, size_t b) { spans.push_back({ kind, a, b }); }
    int collect();                                 // behind the call's last synchronisation
};

// Two receivers that share one device (the lanes of dtl_rx_submit) take turns at the decoder: its 256 persistent workgroups
// fill every CU's LDS, so a second decoder launched next to it only waits inside the device for the first one's workgroups to
// leave -- and its measured duration would count that wait.  The later launch is made to wait for the earlier one's end
// explicitly (an event dependency in front of its own timing events).
struct DecoderGate {
    std::mutex mu;
    hipEvent_t last_done = nullptr;          // end of the decoder launched last (owned by the receiver that recorded it)
};

class Receiver {
public:
    RxConfig cfg;
    DecoderGate* decoder_gate = nullptr;     // set by the owner of several receivers on one device (capi.cpp: the lanes)
    Receiver();
    ~Receiver();
    int init(); // validate, derive tables, parse codes (no GPU needed)
    // one self-contained batch: the stream starts at d_iq[0] and ends at d_iq[n-1]
    int process(const cf32* d_iq, int64_t n, unsigned flags, hipStream_t stream);
    // streaming form: consecutive calls continue one sample stream (what a GNU Radio block's work()
    // sees).  Detector / oscillator / demux / transport-block / feedback state and the samples of
    // frames that are not complete yet are carried to the next call; `final` flushes the stream the
    // way process() treats the end of its batch and resets the state.
    int process_stream(const cf32* d_iq, int64_t n, unsigned flags, bool final, hipStream_t stream);
    void reset_stream() noexcept;
    int64_t stream_position() const; // samples consumed since the last reset
    // device time of the phase-B kernels and of the LDPC kernel(s) of the call `back` calls ago (0 = the last one), from
    // HIP events recorded on that call's stream; -1 where the call had none
    int kernel_times(int back, double* ldpc_ms, double* payload_ms);
    const KernelClock& kernel_clock() const { return kc_; }

    // results of the last process() call
    std::vector<FrameInfo> frames;   // every trigger the demux visited, in order
    std::vector<TbInfo> tbs;
    std::vector<int64_t> raw_triggers, fixed_triggers;
    std::vector<float> trigger_angles; // per raw trigger
    int64_t n_payload_frames = 0;
    int64_t n_payload_bytes = 0;
    int64_t n_llr = 0;
    int64_t n_codewords = 0;
    double stage_ms[8] = { 0 };
    double dbg_plateau_ms = 0;
    unsigned dbg_n_edges = 0;
    DevBuf d_payload, d_eq_hard, d_eq_soft, d_soft_syms, d_llr, d_taps, d_hdr_bits, d_metric, d_cw_bits, d_cw_nit;

    // ---- single blocks of the graph on frequency-domain vectors (C ABI: dtl_chanest, dtl_header_decode,
    // dtl_equalize_frames); stateless, usable next to process()
    int chanest(const cf32* d_sync_syms, int n, int32_t* d_carr_off, cf32* d_taps, hipStream_t st);
    int header_decode(const cf32* d_hdr_syms, const cf32* d_taps, const int32_t* d_carr_off, int n, HeaderRec* recs_host,
                      uint8_t* d_hdr_bits, cf32* d_taps_out, hipStream_t st);
    int equalize_frames(const cf32* d_frames, const cf32* d_taps, const FrameDesc* descs_host, int n, cf32* d_hard,
                        cf32* d_soft, cf32* d_soft_syms, FrameStat* stats_host, hipStream_t st);

    int frame_capacity() const { return capacity_; }
    int frame_samples() const { return (2 + cfg.header_len() + cfg.frame_length) * (cfg.fft_len + cfg.cp_len); }
    int hdr_bits() const { return cfg.header_bits(); }   // == RxTables.hdr_bits: the row stride of d_hdr_bits
    const std::vector<std::unique_ptr<LdpcDecoder>>& decoders() const { return decoders_; }

private:
    struct StreamState;
    struct Call;
    class Worker;
    std::unique_ptr<StreamState> ss_;
    KernelClock kc_;
    std::unique_ptr<Worker> worker_;  // the sync stage's helper thread: started at the first call that needs it, kept
    bool worker_failed_ = false;      // ... it could not be started: the sync stage runs on the caller's thread alone
    int process_stream_body(const cf32* d_iq, int64_t n, unsigned flags, bool final, hipStream_t stream);
    int stream_begin(Call& c);
    int sync_launch(Call& c);
    int stage_sync(Call& c);
    int stage_triggers(Call& c);
    int submit_angles(Call& c, size_t first, size_t count, hipStream_t on);
    int collect_angles(Call& c, size_t parts);
    int speculate_headers(Call& c, int64_t limit, bool split_last = false);
    int stage_frames(Call& c);
    int stage_blocks(Call& c);
    int stage_records(Call& c);
    int stream_end(Call& c);
    int upload_tables();
    RxTables tab_{};
    DevBuf tab_blob_;
    bool uploaded_ = false;
    // HIP events around the payload / LDPC kernels: recorded on the launch stream in every call, in a ring of the last
    // kEvRing calls; read on request (kernel_times), not inside the call -- a caller that times K steps asks afterwards
    static constexpr int kEvRing = 32;
    hipEvent_t ev_ring_[kEvRing][4] = {};
    unsigned ev_mask_[kEvRing] = {};   // bit 0: payload pair recorded, bit 1: decoder pair recorded
    uint64_t n_calls_ = 0;
    hipEvent_t* ev_ = nullptr;         // this call's four events
    int capacity_ = 0;
    std::vector<std::unique_ptr<LdpcDecoder>> decoders_; // index 0 == nullptr ("no FEC")
    DevBuf d_edges_, d_counter_, d_trig_, d_angle_, d_cands_, d_segs_, d_recs_, d_frames_, d_stats_,
        d_cwdesc_, d_tbgather_, d_llrcopy_, d_tbres_, d_len_, d_offs_, d_total_, d_staging_, d_nofec_info_;
    HostBuf h_small_, h_edges_, h_trig_, h_angle_, h_cands_, h_segs_, h_recs_, h_frames_, h_gather_, h_llrcopy_,
        h_stats_, h_tbres_, h_nofec_;
    DevBuf d_work_, d_carry_, d_llr_carry_;
    std::vector<hipEvent_t> chunk_ev_; // one per candidate chunk of the phase A/B pipeline
    hipEvent_t ev_stats_ = nullptr;    // the per-frame SNR estimates have reached the host
    hipEvent_t ev_pre_ = nullptr;      // the decoder's preparations on the side stream (stage_blocks)
    hipEvent_t ev_dec_[2] = { nullptr, nullptr };   // ends of this receiver's last two decoder launches (DecoderGate)
    unsigned dec_launches_ = 0;
    hipStream_t side_ = nullptr;       // arg(P) at the triggers of one sample range while the sync kernel runs on the next
    std::vector<hipEvent_t> angle_ev_; // one per submitted group of triggers
    std::vector<hipEvent_t> hdr_ev_;   // one per header-kernel launch of a call
    // (the same for header candidates and trigger lists, DTL_RX_ZERO_COPY_CANDS / _TRIG: measured, off -- one host read per
    // SHORT workgroup is bound by the rate of small reads over the link: the header kernels took 2.25 instead of 1.0 ms)
    bool zero_copy_cands_ = false, zero_copy_trig_ = false;
    bool zero_copy_descs_ = true;  // phase B reads the frame descriptors from pinned host memory (DTL_RX_ZERO_COPY_DESCS=0: copies)
    int flush_first_ = 2048, flush_frames_ = 8192; // phase B is handed over after this many frames of the demux walk (first / later hand-overs)
    int64_t split_min_ = int64_t(1) << 26; // calls with at least this many new samples run the sync stage in several ranges
    int sync_ranges_ = 4;                  // measured: 4 -> 17.7 ms per C3 step, 6 -> 17.8, 8 -> 18.1 (per-range launch and copy overhead)
};

} // namespace dtl
