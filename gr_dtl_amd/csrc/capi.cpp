// capi.cpp -- extern "C" boundary of libdtl_hip.so (declared in include/dtl_rx.h).
#include "../../include/dtl_rx.h"

#include "dtl_common.h"
#include "dtl_guard.h"
#include "ldpc_bp.h"
#include "rx_kernels.h"
#include "rx_pipeline.h"
#include "tx_pipeline.h"

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <fstream>
#include <memory>
#include <mutex>
#include <new>
#include <sstream>
#include <stdexcept>
#include <system_error>
#include <thread>

#define DTL_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != DTL_OK) return _rc; \
    } while (0)

namespace dtl {
namespace {
thread_local char g_last_error[1024] = "";
}
void set_last_error(const char* msg) noexcept
{
    std::snprintf(g_last_error, sizeof(g_last_error), "%s", msg ? msg : "");
}
void set_last_error(const std::string& msg) noexcept { set_last_error(msg.c_str()); }
const char* last_error() noexcept { return g_last_error; }

} // namespace dtl

struct dtl_ldpc {
    dtl::LdpcDecoder dec;
};

// Batches handed over ahead (dtl_rx_submit / dtl_rx_wait) run on two LANES: each a receiver of its own (lane 0: the handle's,
// lane 1: a second one with the same configuration, made when first needed) on its own host thread and its own stream.
// A receiver's step leaves the device idle where its host is in the loop (the round trips at its front, launch gaps, the
// decoder's tail); the other lane's batch fills exactly those places -- what two handles on two threads do (measured:
// 15.2 -> 13.4 ms per 64 k-frame batch), behind ONE handle and one calling thread.
struct RxLane {
    dtl::Receiver* r = nullptr;
    std::unique_ptr<dtl::Receiver> own;     // lane 1's receiver
    hipStream_t st = nullptr;
    hipEvent_t ev_in = nullptr;              // the caller's stream at submit: the lane's work starts behind it
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, done = false, quit = false, busy = false;
    const dtl::cf32* iq = nullptr;
    int64_t n = 0;
    unsigned flags = 0;
    int device = 0;                          // the device of the lane's stream and buffers
    int rc = DTL_OK;
    char err[512] = "";
    void loop()
    {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return has_job || quit; });
                if (quit) return;
                has_job = false;
            }
            int code = ::dtl::guarded("dtl_rx_submit (lane)", [&]() -> int {
                // (a new thread starts on device 0: the batch runs on the device that was current for the caller at submit --
                // one process per GPU sets it per rank -- unless the handle is bound to one, dtl_rx_cfg.device)
                DTL_HIP_CHECK(hipSetDevice(device));
                DTL_HIP_CHECK(hipStreamWaitEvent(st, ev_in, 0));
                return r->process(iq, n, flags, st);
            });
            {
                std::lock_guard<std::mutex> lk(mu);
                rc = code;
                std::snprintf(err, sizeof(err), "%s", code == DTL_OK ? "" : ::dtl::last_error());   // (the text is per thread)
                done = true;
            }
            cv.notify_all();
        }
    }
    int finish()      // waits for the lane's batch; its status
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        done = false;
        busy = false;
        return rc;
    }
    ~RxLane()
    {
        if (th.joinable()) {
            if (busy) (void)finish();
            { std::lock_guard<std::mutex> lk(mu); quit = true; }
            cv.notify_all();
            th.join();
        }
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    }
};

struct dtl_rx {
    dtl::DecoderGate gate;                    // the lanes' decoders take turns on the device (declared first: outlives the receivers)
    dtl::Receiver rx;
    // dtl_rx_submit / dtl_rx_wait
    std::unique_ptr<RxLane> lane[2];
    std::deque<int> waiting;                  // lanes of the batches handed over, oldest first
    unsigned n_submitted = 0;
    std::deque<int> recent;                   // lanes of the last calls that returned results, newest first (kernel-time queries)
    dtl::Receiver* last = nullptr;            // the receiver of the last call that returned results
    dtl::Receiver& current() { return last ? *last : rx; }
    void returned(int lane_no, dtl::Receiver* r)
    {
        last = r;
        recent.push_front(lane_no);
        if (recent.size() > 64) recent.pop_back();
    }
    ~dtl_rx() { lane[1].reset(); lane[0].reset(); }   // (the lanes' threads end before the receivers they drive)
};

struct dtl_tx {
    dtl::Transmitter tx;
};

static_assert(sizeof(dtl_cw_desc) == sizeof(dtl::CwDesc), "C ABI gather descriptor layout");
static_assert(sizeof(dtl_rx_frame_info) == sizeof(dtl::FrameInfo), "C ABI frame info layout");
static_assert(sizeof(dtl_rx_tb_info) == sizeof(dtl::TbInfo), "C ABI TB info layout");
static_assert(sizeof(dtl_tx_frame_info) == sizeof(dtl::TxFrameInfo), "C ABI TX frame info layout");
static_assert(sizeof(dtl_header_fields) == sizeof(dtl::HeaderRec), "C ABI header fields layout");
static_assert(sizeof(dtl_frame_eq_out) == sizeof(dtl::FrameStat), "C ABI frame equaliser output layout");
static_assert(sizeof(dtl_tb_span) == sizeof(dtl::TbSpan), "C ABI transport-block span layout");

extern "C" {

const char* dtl_last_error(void) { return dtl::last_error(); }
const char* dtl_version(void) { return "dtl-mi355x 0.1 (gfx950)"; }

int dtl_device_count(void)
{
    DTL_API_BEGIN("dtl_device_count")
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
    DTL_API_END
}

int dtl_ldpc_create_from_text(const char* text, size_t len, int max_iterations, float sigma,
                              dtl_ldpc** out)
{
    DTL_API_BEGIN("dtl_ldpc_create_from_text")
    if (!text || !out) { dtl::set_last_error("ldpc_create: null argument"); return DTL_ERR_INVALID_ARG; }
    *out = nullptr;
    auto h = std::make_unique<dtl_ldpc>();    // (owned until the handle is complete: an exception below must not leak it)
    std::string err = h->dec.code.parse_alist(std::string(text, len));
    if (!err.empty()) { dtl::set_last_error(err); return DTL_ERR_INVALID_ARG; }
    h->dec.code.reduce();
    h->dec.code.build_schedules();
    if (max_iterations > 0) h->dec.max_iterations = max_iterations;
    if (sigma > 0) h->dec.sigma = sigma;
    *out = h.release();
    return DTL_OK;
    DTL_API_END
}

int dtl_ldpc_create(const char* path, int max_iterations, float sigma, dtl_ldpc** out)
{
    DTL_API_BEGIN("dtl_ldpc_create")
    if (!path || !out) { dtl::set_last_error("ldpc_create: null argument"); return DTL_ERR_INVALID_ARG; }
    std::ifstream f(path);
    if (!f) { dtl::set_last_error(std::string("cannot open alist ") + path); return DTL_ERR_IO; }
    std::stringstream ss;
    ss << f.rdbuf();
    std::string text = ss.str();
    return dtl_ldpc_create_from_text(text.data(), text.size(), max_iterations, sigma, out);
    DTL_API_END
}

void dtl_ldpc_destroy(dtl_ldpc* h) { delete h; }

int dtl_ldpc_set_arith(dtl_ldpc* h, int32_t arith)
{
    DTL_API_BEGIN("dtl_ldpc_set_arith")
    if (!h || (arith != DTL_ARITH_GR_EXACT && arith != DTL_ARITH_FAST)) { dtl::set_last_error("ldpc_set_arith: bad argument"); return DTL_ERR_INVALID_ARG; }
    h->dec.arith = arith;
    return DTL_OK;
    DTL_API_END
}
int dtl_ldpc_get_n(const dtl_ldpc* h) { return h ? h->dec.code.n : DTL_ERR_INVALID_ARG; }
int dtl_ldpc_get_k(const dtl_ldpc* h) { return h ? h->dec.code.k : DTL_ERR_INVALID_ARG; }

int dtl_ldpc_get_permute(const dtl_ldpc* h, int32_t* out)
{
    DTL_API_BEGIN("dtl_ldpc_get_permute")
    if (!h || !out) return DTL_ERR_INVALID_ARG;
    for (int i = 0; i < h->dec.code.n; i++) out[i] = h->dec.code.permute[i];
    return DTL_OK;
    DTL_API_END
}

int dtl_ldpc_encode_host(const dtl_ldpc* h, const uint8_t* data, uint8_t* out)
{
    DTL_API_BEGIN("dtl_ldpc_encode_host")
    if (!h || !data || !out) return DTL_ERR_INVALID_ARG;
    h->dec.code.encode(data, out);
    return DTL_OK;
    DTL_API_END
}

int dtl_ldpc_encode_batch_host(const dtl_ldpc* h, const uint8_t* data, size_t ncw, uint8_t* out)
{
    DTL_API_BEGIN("dtl_ldpc_encode_batch_host")
    if (!h || !data || !out) return DTL_ERR_INVALID_ARG;
    const int n = h->dec.code.n, k = h->dec.code.k;
    for (size_t i = 0; i < ncw; i++) h->dec.code.encode(data + i * k, out + i * n);
    return DTL_OK;
    DTL_API_END
}

// device tables are created lazily so that creating/inspecting a code needs no GPU
static int ensure_uploaded(const dtl_ldpc* h)
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    auto* d = const_cast<dtl::LdpcDecoder*>(&h->dec);
    if (d->uploaded()) return DTL_OK;
    if (d->lds_bytes() == 0) return DTL_ERR_INVALID_ARG;
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device: the HIP path has no CPU fallback"); return DTL_ERR_NO_DEVICE; }
    return d->upload();
}

int dtl_ldpc_decode_gather(const dtl_ldpc* h, const float* d_llr, const dtl_cw_desc* d_desc,
                           size_t ncw, uint8_t* d_bits, int32_t* d_nit, void* stream)
{
    DTL_API_BEGIN("dtl_ldpc_decode_gather")
    if (!h) { dtl::set_last_error("ldpc_decode: null handle"); return DTL_ERR_INVALID_ARG; }
    if (!h->dec.uploaded()) {
        int rc = ensure_uploaded(h);
        if (rc != DTL_OK) return rc;
    }
    return h->dec.decode(d_llr, reinterpret_cast<const dtl::CwDesc*>(d_desc), ncw, d_bits, d_nit,
                         static_cast<hipStream_t>(stream));
    DTL_API_END
}

int dtl_ldpc_decode_batch(const dtl_ldpc* h, const float* d_llr, size_t ncw, uint8_t* d_bits,
                          int32_t* d_nit, void* stream)
{
    DTL_API_BEGIN("dtl_ldpc_decode_batch")
    return dtl_ldpc_decode_gather(h, d_llr, nullptr, ncw, d_bits, d_nit, stream);
    DTL_API_END
}


static int fill_config(const dtl_rx_cfg* c, dtl::RxConfig& g, const char* who)
{
    if (c->fft_len <= 0 || c->n_occ_sets <= 0 || !c->occ || !c->occ_len || !c->sync_word1 || !c->sync_word2) {
        dtl::set_last_error(std::string(who) + ": incomplete configuration");
        return DTL_ERR_INVALID_ARG;
    }
    // The shape of the flattened arrays, before anything is read through them (what the values MEAN -- carrier indices,
    // set sizes, code tables -- is checked by init() on the copies): every count bounds a pointer walk below.
    auto refuse = [&](const char* what) {
        dtl::set_last_error(std::string(who) + ": " + what);
        return DTL_ERR_INVALID_ARG;
    };
    if (c->fft_len > 65536) return refuse("fft_len out of range");
    if (c->cp_len < 0 || c->cp_len > c->fft_len) return refuse("bad cp_len");
    if (c->frame_length < 1 || c->frame_length > 65535) return refuse("frame_length must lie in [1, 65535]");
    if (c->n_occ_sets > 4096 || c->max_occ < 1 || c->max_occ > c->fft_len) return refuse("occupied carrier sets: bad shape");
    for (int s = 0; s < c->n_occ_sets; s++)
        if (c->occ_len[s] < 0 || c->occ_len[s] > c->max_occ) return refuse("occupied carrier set longer than max_occ (or negative)");
    if (c->n_pilot_sets < 0 || c->n_pilot_sets > 4096 || c->n_pilots < 0 || c->n_pilots > c->fft_len) return refuse("pilot carrier sets: bad shape");
    if (c->n_pilot_sets > 0 && c->n_pilots > 0 && !c->pilot_carriers) return refuse("pilot_carriers is null");
    if (c->n_pilot_sym_sets < 0 || c->n_pilot_sym_sets > 65536) return refuse("pilot symbol sets: bad count");
    if (c->n_pilot_sym_sets > 0 && c->n_pilots > 0 && !c->pilot_symbols) return refuse("pilot_symbols is null");
    if (c->n_fec_codes < 0 || c->n_fec_codes > 255) return refuse("n_fec_codes must lie in [0, 255]");
    if (c->n_fec_codes > 0 && (!c->alist_texts || !c->alist_lens)) return refuse("code tables: null array");
    for (int i = 0; i < c->n_fec_codes; i++)
        if (!c->alist_texts[i] || c->alist_lens[i] > (size_t(1) << 28)) return refuse("code table: null text or absurd length");
    if (c->n_mcs < 0 || c->n_mcs > 4096) return refuse("n_mcs must lie in [0, 4096]");
    if (c->n_mcs > 0 && (!c->mcs_snr || !c->mcs_cnst || !c->mcs_fec)) return refuse("MCS table: null array");
    for (int i = 0; i < c->n_mcs; i++) {
        if (c->mcs_cnst[i] < 1 || c->mcs_cnst[i] > 4) return refuse("MCS table: Unknown constellation");   // constellation_type_t
        // (an index without a table is legal here: a handle may serve the single-block entry points only, and process()
        // answers "Misconfiguration: decoder not found!" like the reference's decoder block)
        if (c->mcs_fec[i] < 0 || c->mcs_fec[i] > 255) return refuse("MCS table: code index out of range");
    }
    if (c->n_mcs > 0 && (c->initial_mcs < 0 || c->initial_mcs >= c->n_mcs)) return refuse("initial_mcs outside the MCS table");
    if (c->max_iterations < 0 || c->max_iterations > 100000) return refuse("max_iterations out of range");
    if (!(c->sync_threshold == c->sync_threshold)) return refuse("sync_threshold is NaN");
    if (!(c->eq_alpha >= 0.0f && c->eq_alpha <= 1.0f)) return refuse("eq_alpha must lie in [0, 1]");
    if (!(c->bp_sigma >= 0.0f) || !(c->bp_sigma < 1.0e18f)) return refuse("bp_sigma must be a finite number >= 0");
    if (!(c->hysteresis == c->hysteresis)) return refuse("hysteresis is NaN");
    g.fft_len = c->fft_len; g.cp_len = c->cp_len; g.frame_length = c->frame_length; g.has_fec = c->has_fec != 0;
    for (int s = 0; s < c->n_occ_sets; s++)
        g.occupied_carriers.emplace_back(c->occ + size_t(s) * c->max_occ, c->occ + size_t(s) * c->max_occ + c->occ_len[s]);
    for (int s = 0; s < c->n_pilot_sets; s++)
        g.pilot_carriers.emplace_back(c->pilot_carriers + size_t(s) * c->n_pilots, c->pilot_carriers + size_t(s + 1) * c->n_pilots);
    for (int s = 0; s < c->n_pilot_sym_sets; s++) {
        std::vector<dtl::cf32> row(c->n_pilots);
        for (int k = 0; k < c->n_pilots; k++) row[k] = { c->pilot_symbols[2 * (size_t(s) * c->n_pilots + k)], c->pilot_symbols[2 * (size_t(s) * c->n_pilots + k) + 1] };
        g.pilot_symbols.push_back(std::move(row));
    }
    g.sync_word1.resize(c->fft_len); g.sync_word2.resize(c->fft_len);
    for (int i = 0; i < c->fft_len; i++) {
        g.sync_word1[i] = { c->sync_word1[2 * i], c->sync_word1[2 * i + 1] };
        g.sync_word2[i] = { c->sync_word2[2 * i], c->sync_word2[2 * i + 1] };
    }
    g.sync_threshold = c->sync_threshold; g.eq_alpha = c->eq_alpha;
    g.use_sync_correct = c->use_sync_correct != 0; g.scramble_header = c->scramble_header != 0;
    g.llr_dist_mode = c->llr_dist_mode; g.max_iterations = c->max_iterations; g.bp_sigma = c->bp_sigma;
    for (int i = 0; i < c->n_fec_codes; i++) g.alist_texts.emplace_back(c->alist_texts[i], c->alist_lens[i]);
    for (int i = 0; i < c->n_mcs; i++) g.mcs.push_back({ c->mcs_snr[i], c->mcs_cnst[i], c->mcs_fec[i] });
    g.initial_mcs = c->initial_mcs; g.hysteresis = c->hysteresis; g.decision_th = c->decision_th > 0 ? c->decision_th : 5;
    if (c->arith != DTL_ARITH_GR_EXACT && c->arith != DTL_ARITH_FAST) { dtl::set_last_error(std::string(who) + ": arith must be DTL_ARITH_GR_EXACT or DTL_ARITH_FAST"); return DTL_ERR_INVALID_ARG; }
    g.arith = c->arith;
    if (c->device < -1) { dtl::set_last_error(std::string(who) + ": device must be a HIP device index or DTL_DEVICE_CURRENT"); return DTL_ERR_INVALID_ARG; }
    g.device = c->device;
    return DTL_OK;
}

int dtl_rx_create(const dtl_rx_cfg* c, dtl_rx** out)
{
    DTL_API_BEGIN("dtl_rx_create")
    if (!c || !out) { dtl::set_last_error("rx_create: null argument"); return DTL_ERR_INVALID_ARG; }
    *out = nullptr;
    auto h = std::make_unique<dtl_rx>();
    int rc = fill_config(c, h->rx.cfg, "rx_create");
    if (rc == DTL_OK) rc = h->rx.init();
    if (rc != DTL_OK) return rc;
    *out = h.release();
    return DTL_OK;
    DTL_API_END
}

int dtl_tx_create(const dtl_rx_cfg* c, dtl_tx** out)
{
    DTL_API_BEGIN("dtl_tx_create")
    if (!c || !out) { dtl::set_last_error("tx_create: null argument"); return DTL_ERR_INVALID_ARG; }
    *out = nullptr;
    auto h = std::make_unique<dtl_tx>();
    int rc = fill_config(c, h->tx.cfg, "tx_create");
    if (rc == DTL_OK) rc = h->tx.init();
    if (rc != DTL_OK) return rc;
    *out = h.release();
    return DTL_OK;
    DTL_API_END
}

void dtl_tx_destroy(dtl_tx* h) { delete h; }

int dtl_constellation_points_host(int32_t constellation, float* points16, int32_t* n_points)
{
    DTL_API_BEGIN("dtl_constellation_points_host")
    if (!points16 || !n_points || constellation < 1 || constellation > 4) { dtl::set_last_error("Unknown constellation"); return DTL_ERR_INVALID_ARG; }
    dtl::cf32 p[16];
    dtl::constellation_points_host(constellation, p);
    std::memcpy(points16, p, sizeof(p));
    *n_points = 1 << constellation;
    return DTL_OK;
    DTL_API_END
}

int dtl_header_format_host(const dtl_tx_frame_info* fields, int32_t has_fec, int32_t header_len_bits, uint8_t* bits_out)
{
    DTL_API_BEGIN("dtl_header_format_host")
    if (!fields || !bits_out || header_len_bits < 16) { dtl::set_last_error("header_format: bad argument"); return DTL_ERR_INVALID_ARG; }
    dtl::TxFrameInfo f;
    std::memcpy(&f, fields, sizeof(f));
    dtl::format_header_bits(f, has_fec != 0, header_len_bits, bits_out);
    return DTL_OK;
    DTL_API_END
}

int dtl_tx_plan(dtl_tx* h, int64_t n_user, int32_t constellation, int32_t fec, int32_t first_frame_no,
                int32_t first_tb_no, dtl_tx_frame_info* frames_out, int64_t cap, int64_t* n_out)
{
    DTL_API_BEGIN("dtl_tx_plan")
    if (!h || !n_out || n_user < 0 || cap < 0 || (cap && !frames_out)) { dtl::set_last_error("tx_plan: bad argument"); return DTL_ERR_INVALID_ARG; }
    int rc = h->tx.plan(n_user, constellation, fec, first_frame_no, first_tb_no);
    if (rc != DTL_OK) return rc;
    *n_out = int64_t(h->tx.frames.size());
    if (*n_out > cap) { dtl::set_last_error("tx_plan: frame capacity too small"); return DTL_ERR_CAPACITY; }
    std::memcpy(frames_out, h->tx.frames.data(), h->tx.frames.size() * sizeof(dtl_tx_frame_info));
    return DTL_OK;
    DTL_API_END
}

int dtl_tx_process(dtl_tx* h, const uint8_t* d_user, int64_t n_user, int32_t constellation, int32_t fec,
                   int32_t first_frame_no, int32_t first_tb_no, int32_t gap, int32_t want_syms, void* stream,
                   dtl_tx_result* out)
{
    DTL_API_BEGIN("dtl_tx_process")
    if (!h || !out) { dtl::set_last_error("tx_process: null argument"); return DTL_ERR_INVALID_ARG; }
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device: the HIP path has no CPU fallback"); return DTL_ERR_NO_DEVICE; }
    dtl::Transmitter& t = h->tx;
    int rc = t.process(d_user, n_user, constellation, fec, first_frame_no, first_tb_no, gap, want_syms != 0,
                       static_cast<hipStream_t>(stream));
    if (rc != DTL_OK) return rc;
    std::memset(out, 0, sizeof(*out));
    out->n_frames = int64_t(t.frames.size());
    out->n_samples = t.n_samples;
    out->n_tbs = t.n_tbs;
    out->frames = reinterpret_cast<const dtl_tx_frame_info*>(t.frames.data());
    out->d_iq = t.d_iq.as<float>();
    out->d_syms = want_syms ? t.d_syms.as<uint8_t>() : nullptr;
    out->d_hdr_bits = t.d_hdr_bits.as<uint8_t>();
    out->frame_capacity = t.frame_capacity();
    out->hdr_bits = t.hdr_bits();
    out->frame_samples = t.frame_samples();
    return DTL_OK;
    DTL_API_END
}

void dtl_rx_destroy(dtl_rx* h) { delete h; }

static void fill_result(dtl::Receiver& r, uint32_t flags, dtl_rx_result* out)
{
    std::memset(out, 0, sizeof(*out));
    out->n_frames = int64_t(r.frames.size());
    out->n_payload_frames = r.n_payload_frames;
    out->n_tbs = int64_t(r.tbs.size());
    out->n_payload_bytes = r.n_payload_bytes;
    out->n_raw_triggers = int64_t(r.raw_triggers.size());
    out->n_fixed_triggers = int64_t(r.fixed_triggers.size());
    out->n_llr = r.n_llr; out->n_codewords = r.n_codewords;
    out->frames = reinterpret_cast<const dtl_rx_frame_info*>(r.frames.data());
    out->tbs = reinterpret_cast<const dtl_rx_tb_info*>(r.tbs.data());
    out->raw_triggers = r.raw_triggers.data();
    out->fixed_triggers = r.fixed_triggers.data();
    out->trigger_angles = r.trigger_angles.data();
    out->d_payload = r.d_payload.as<uint8_t>();
    out->d_eq_hard = (flags & DTL_RX_WANT_EQ_HARD) ? r.d_eq_hard.as<float>() : nullptr;
    out->d_eq_soft = (flags & DTL_RX_WANT_EQ_SOFT) ? r.d_eq_soft.as<float>() : nullptr;
    out->d_soft_syms = (flags & DTL_RX_WANT_SOFT_SYMS) ? r.d_soft_syms.as<float>() : nullptr;
    out->d_llr = r.d_llr.as<float>();
    out->d_taps = r.d_taps.as<float>();
    out->d_hdr_bits = (flags & DTL_RX_WANT_HDR_BITS) ? r.d_hdr_bits.as<uint8_t>() : nullptr;
    out->d_metric = (flags & DTL_RX_WANT_METRIC) ? r.d_metric.as<float>() : nullptr;
    out->d_cw_bits = r.d_cw_bits.as<uint8_t>();
    out->d_cw_nit = r.d_cw_nit.as<int32_t>();
    for (int i = 0; i < 8; i++) out->stage_ms[i] = r.stage_ms[i];
    out->frame_capacity = r.frame_capacity();
    out->hdr_bits = r.hdr_bits();            // the row stride of d_hdr_bits (RxTables.hdr_bits)
    out->frame_samples = r.frame_samples();
}

static int rx_run(dtl_rx* h, const float* d_iq, int64_t n, uint32_t flags, int mode, void* stream, dtl_rx_result* out)
{
    if (!h || !out) { dtl::set_last_error("rx_process: null argument"); return DTL_ERR_INVALID_ARG; }
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device: the HIP path has no CPU fallback"); return DTL_ERR_NO_DEVICE; }
    if (!h->waiting.empty()) { dtl::set_last_error("rx_process: batches of dtl_rx_submit are waiting (dtl_rx_wait first)"); return DTL_ERR_INVALID_ARG; }
    dtl::Receiver& r = h->rx;
    const auto* iq = reinterpret_cast<const dtl::cf32*>(d_iq);
    auto* st = static_cast<hipStream_t>(stream);
    int rc = mode < 0 ? r.process(iq, n, flags, st) : r.process_stream(iq, n, flags, mode != 0, st);
    if (rc != DTL_OK) return rc;
    h->returned(0, &h->rx);
    fill_result(r, flags, out);
    return DTL_OK;
}

int dtl_rx_submit(dtl_rx* h, const float* d_iq, int64_t n, uint32_t flags, void* stream)
{
    DTL_API_BEGIN("dtl_rx_submit")
    if (!h || n < 0 || (n > 0 && !d_iq)) { dtl::set_last_error("rx_submit: bad argument"); return DTL_ERR_INVALID_ARG; }
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device: the HIP path has no CPU fallback"); return DTL_ERR_NO_DEVICE; }
    if (h->waiting.size() >= 2) { dtl::set_last_error("submit: two batches are waiting already (dtl_rx_wait takes the oldest)"); return DTL_ERR_CAPACITY; }
    int ln = int(h->n_submitted & 1u);
    if (h->lane[ln] && h->lane[ln]->busy) ln ^= 1;                // (fewer than two are waiting: the other lane is free)
    dtl::DeviceScope dev;
    DTL_TRY(dev.enter(h->rx.cfg.device));
    int device = 0;
    DTL_HIP_CHECK(hipGetDevice(&device));                         // the handle's device, or the caller's current one
    if (h->lane[ln] && h->lane[ln]->device != device) {
        dtl::set_last_error("submit: the handle's lanes were made on another device (bind the handle: dtl_rx_cfg.device)");
        return DTL_ERR_INVALID_ARG;
    }
    if (!h->lane[ln]) {
        auto L = std::make_unique<RxLane>();
        L->device = device;
        if (ln == 0) { L->r = &h->rx; h->rx.decoder_gate = &h->gate; }
        else {
            L->own = std::make_unique<dtl::Receiver>();
            L->own->cfg = h->rx.cfg;
            L->own->decoder_gate = &h->gate;
            DTL_TRY(L->own->init());
            L->r = L->own.get();
        }
        DTL_HIP_CHECK(hipStreamCreateWithFlags(&L->st, hipStreamNonBlocking));
        DTL_HIP_CHECK(hipEventCreateWithFlags(&L->ev_in, hipEventDisableTiming));
        RxLane* raw = L.get();
        L->th = std::thread([raw] { raw->loop(); });
        h->lane[ln] = std::move(L);
    }
    RxLane& L = *h->lane[ln];
    DTL_HIP_CHECK(hipEventRecord(L.ev_in, static_cast<hipStream_t>(stream)));
    {
        std::lock_guard<std::mutex> lk(L.mu);
        L.iq = reinterpret_cast<const dtl::cf32*>(d_iq); L.n = n; L.flags = flags;
        L.has_job = true; L.done = false; L.busy = true;
    }
    L.cv.notify_all();
    h->waiting.push_back(ln);
    h->n_submitted++;
    return DTL_OK;
    DTL_API_END
}

int dtl_rx_wait(dtl_rx* h, dtl_rx_result* out)
{
    DTL_API_BEGIN("dtl_rx_wait")
    if (!h || !out) { dtl::set_last_error("rx_wait: null argument"); return DTL_ERR_INVALID_ARG; }
    if (h->waiting.empty()) { dtl::set_last_error("wait: no batch is waiting (dtl_rx_submit first)"); return DTL_ERR_INVALID_ARG; }
    const int ln = h->waiting.front();
    h->waiting.pop_front();
    RxLane& L = *h->lane[ln];
    const int rc = L.finish();
    if (rc != DTL_OK) { dtl::set_last_error(L.err); return rc; }
    h->returned(ln, L.r);
    fill_result(*L.r, L.flags, out);
    return DTL_OK;
    DTL_API_END
}

int dtl_rx_process(dtl_rx* h, const float* d_iq, int64_t n, uint32_t flags, void* stream, dtl_rx_result* out)
{
    DTL_API_BEGIN("dtl_rx_process")
    return rx_run(h, d_iq, n, flags, -1, stream, out);
    DTL_API_END
}

int dtl_rx_process_stream(dtl_rx* h, const float* d_iq, int64_t n, uint32_t flags, int32_t final, void* stream,
                          dtl_rx_result* out)
{
    DTL_API_BEGIN("dtl_rx_process_stream")
    return rx_run(h, d_iq, n, flags, final ? 1 : 0, stream, out);
    DTL_API_END
}

void dtl_rx_reset(dtl_rx* h) { if (h) h->rx.reset_stream(); }

int64_t dtl_rx_stream_position(const dtl_rx* h) { return h ? h->rx.stream_position() : 0; }

int dtl_rx_kernel_ms(dtl_rx* h, int32_t back, double* ldpc_ms, double* payload_ms)
{
    DTL_API_BEGIN("dtl_rx_kernel_ms")
    if (!h) { dtl::set_last_error("rx_kernel_ms: null handle"); return DTL_ERR_INVALID_ARG; }
    // call `back` calls ago: which lane served it, and how many of that lane's calls are younger
    if (back < 0 || size_t(back) >= h->recent.size()) return h->rx.kernel_times(back, ldpc_ms, payload_ms);
    const int ln = h->recent[size_t(back)];
    int younger = 0;
    for (int j = 0; j < back; j++) younger += h->recent[size_t(j)] == ln;
    dtl::Receiver* r = ln == 1 && h->lane[1] ? h->lane[1]->r : &h->rx;
    return r->kernel_times(younger, ldpc_ms, payload_ms);
    DTL_API_END
}

int dtl_rx_kernel_split(dtl_rx* h, dtl_rx_kernel_times* out)
{
    DTL_API_BEGIN("dtl_rx_kernel_split")
    if (!h || !out) { dtl::set_last_error("rx_kernel_split: null argument"); return DTL_ERR_INVALID_ARG; }
    const dtl::KernelClock& kc = h->current().kernel_clock();
    if (!kc.valid) { dtl::set_last_error("rx_kernel_split: the last call was not made with DTL_RX_WANT_KERNEL_SPLIT"); return DTL_ERR_INVALID_ARG; }
    static_assert(int(dtl::KernelClock::N_KINDS) == int(DTL_K_COUNT), "kernel ids of the C ABI");
    for (int k = 0; k < DTL_K_COUNT; k++) { out->ms[k] = kc.ms[k]; out->launches[k] = kc.launches[k]; out->longest_launch_ms[k] = kc.longest[k]; }
    return DTL_OK;
    DTL_API_END
}

int dtl_sync_sc(const float* d_iq, int64_t n, int32_t fft_len, int32_t cp_len, float threshold, float* d_metric,
                int64_t* triggers_host, float* angles_host, int64_t cap, int64_t* n_out, void* stream)
{
    DTL_API_BEGIN("dtl_sync_sc")
    if (!n_out || n < 0 || (n && !d_iq) || cap < 0 || (cap && (!triggers_host || !angles_host))) {
        dtl::set_last_error("sync_sc: bad argument");
        return DTL_ERR_INVALID_ARG;
    }
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device: the HIP path has no CPU fallback"); return DTL_ERR_NO_DEVICE; }
    *n_out = 0;
    if (n == 0) return DTL_OK;
    if (n > int64_t(0x7fffffff)) { dtl::set_last_error("sync_sc: at most 2^31-1 samples per call"); return DTL_ERR_INVALID_ARG; }
    auto* st = static_cast<hipStream_t>(stream);
    const auto* iq = reinterpret_cast<const dtl::cf32*>(d_iq);
    dtl::DevBuf d_edges, d_cnt, d_trig, d_ang;
    const unsigned max_edges = unsigned(std::min<int64_t>(std::max<int64_t>(n / 8, 1024), 1 << 26));
    int rc = d_edges.ensure(size_t(max_edges) * 8);
    if (rc == DTL_OK) rc = d_cnt.ensure(16);
    if (rc != DTL_OK) return rc;
    DTL_HIP_CHECK(hipMemsetAsync(d_cnt.p, 0, 16, st));
    rc = dtl::launch_sync_edges(iq, n, 0, fft_len, threshold, d_edges.as<unsigned long long>(), d_cnt.as<unsigned int>(),
                                max_edges, d_metric, st);
    if (rc != DTL_OK) return rc;
    unsigned n_edges = 0;
    DTL_HIP_CHECK(hipMemcpyAsync(&n_edges, d_cnt.p, 4, hipMemcpyDeviceToHost, st));
    DTL_HIP_CHECK(hipStreamSynchronize(st));
    if (n_edges > max_edges) { dtl::set_last_error("sync: too many threshold crossings (noise-like input?)"); return DTL_ERR_CAPACITY; }
    std::vector<unsigned long long> edges(n_edges);
    if (n_edges) DTL_HIP_CHECK(hipMemcpy(edges.data(), d_edges.p, size_t(n_edges) * 8, hipMemcpyDeviceToHost));
    std::vector<int64_t> trig = dtl::plateau_from_edges(std::move(edges), n, cp_len);
    *n_out = int64_t(trig.size());
    if (int64_t(trig.size()) > cap) { dtl::set_last_error("sync_sc: trigger capacity too small"); return DTL_ERR_CAPACITY; }
    if (trig.empty()) return DTL_OK;
    rc = d_trig.ensure(trig.size() * 8);
    if (rc == DTL_OK) rc = d_ang.ensure(trig.size() * 4);
    if (rc != DTL_OK) return rc;
    DTL_HIP_CHECK(hipMemcpyAsync(d_trig.p, trig.data(), trig.size() * 8, hipMemcpyHostToDevice, st));
    rc = dtl::launch_trigger_angle(iq, n, fft_len, d_trig.as<int64_t>(), int(trig.size()), d_ang.as<float>(), st);
    if (rc != DTL_OK) return rc;
    DTL_HIP_CHECK(hipMemcpyAsync(angles_host, d_ang.p, trig.size() * 4, hipMemcpyDeviceToHost, st));
    DTL_HIP_CHECK(hipStreamSynchronize(st));
    std::copy(trig.begin(), trig.end(), triggers_host);
    return DTL_OK;
    DTL_API_END
}

int dtl_fft_batch(const float* d_in, float* d_out, int64_t n_vectors, int32_t fft_len, int32_t forward, void* stream)
{
    DTL_API_BEGIN("dtl_fft_batch")
    if (n_vectors < 0 || (n_vectors && (!d_in || !d_out))) { dtl::set_last_error("fft_batch: bad argument"); return DTL_ERR_INVALID_ARG; }
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device: the HIP path has no CPU fallback"); return DTL_ERR_NO_DEVICE; }
    return dtl::launch_fft_batch(reinterpret_cast<const dtl::cf32*>(d_in), reinterpret_cast<dtl::cf32*>(d_out), n_vectors,
                                 fft_len, forward, static_cast<hipStream_t>(stream));
    DTL_API_END
}

int dtl_soft_demap(const float* d_syms, int64_t n_syms, int32_t constellation, double noise, int32_t dist_mode,
                   float* d_llr, void* stream)
{
    DTL_API_BEGIN("dtl_soft_demap")
    if (n_syms < 0 || (n_syms && (!d_syms || !d_llr))) { dtl::set_last_error("soft_demap: bad argument"); return DTL_ERR_INVALID_ARG; }
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device: the HIP path has no CPU fallback"); return DTL_ERR_NO_DEVICE; }
    return dtl::launch_soft_demap(reinterpret_cast<const dtl::cf32*>(d_syms), n_syms, constellation, noise, dist_mode, d_llr,
                                  static_cast<hipStream_t>(stream));
    DTL_API_END
}

int dtl_crc32(const uint8_t* d_bytes, int64_t len, uint32_t* crc_host, void* stream)
{
    DTL_API_BEGIN("dtl_crc32")
    if (!crc_host || len < 0 || (len && !d_bytes)) { dtl::set_last_error("crc32: bad argument"); return DTL_ERR_INVALID_ARG; }
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device: the HIP path has no CPU fallback"); return DTL_ERR_NO_DEVICE; }
    auto* st = static_cast<hipStream_t>(stream);
    dtl::DevBuf d_out;
    int rc = d_out.ensure(16);
    if (rc != DTL_OK) return rc;
    rc = dtl::launch_crc32(d_bytes, len, d_out.as<uint32_t>(), st);
    if (rc != DTL_OK) return rc;
    DTL_HIP_CHECK(hipMemcpyAsync(crc_host, d_out.p, 4, hipMemcpyDeviceToHost, st));
    DTL_HIP_CHECK(hipStreamSynchronize(st));
    return DTL_OK;
    DTL_API_END
}

int dtl_chanest(dtl_rx* h, const float* d_sync_syms, int64_t n, int32_t* d_carr_offset, float* d_taps, void* stream)
{
    DTL_API_BEGIN("dtl_chanest")
    if (!h || n < 0 || n > 0x7fffffff) { dtl::set_last_error("chanest: bad argument"); return DTL_ERR_INVALID_ARG; }
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device"); return DTL_ERR_NO_DEVICE; }
    return h->rx.chanest(reinterpret_cast<const dtl::cf32*>(d_sync_syms), int(n), d_carr_offset,
                         reinterpret_cast<dtl::cf32*>(d_taps), static_cast<hipStream_t>(stream));
    DTL_API_END
}

int dtl_header_decode(dtl_rx* h, const float* d_hdr_syms, const float* d_taps, const int32_t* d_carr_offset, int64_t n,
                      dtl_header_fields* fields_host, uint8_t* d_hdr_bits, float* d_taps_out, void* stream)
{
    DTL_API_BEGIN("dtl_header_decode")
    if (!h || n < 0 || n > 0x7fffffff) { dtl::set_last_error("header_decode: bad argument"); return DTL_ERR_INVALID_ARG; }
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device"); return DTL_ERR_NO_DEVICE; }
    return h->rx.header_decode(reinterpret_cast<const dtl::cf32*>(d_hdr_syms), reinterpret_cast<const dtl::cf32*>(d_taps),
                               d_carr_offset, int(n), reinterpret_cast<dtl::HeaderRec*>(fields_host), d_hdr_bits,
                               reinterpret_cast<dtl::cf32*>(d_taps_out), static_cast<hipStream_t>(stream));
    DTL_API_END
}

int dtl_equalize_frames(dtl_rx* h, const float* d_frames, const float* d_taps, const dtl_frame_eq_in* in_host, int64_t n,
                        float* d_eq_hard, float* d_eq_soft, float* d_soft_syms, dtl_frame_eq_out* out_host, void* stream)
{
    DTL_API_BEGIN("dtl_equalize_frames")
    if (!h || n < 0 || n > 0x7fffffff || (n > 0 && !in_host)) { dtl::set_last_error("equalize_frames: bad argument"); return DTL_ERR_INVALID_ARG; }
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device"); return DTL_ERR_NO_DEVICE; }
    std::vector<dtl::FrameDesc> descs(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++) {
        dtl::FrameDesc& d = descs[size_t(i)];
        d = dtl::FrameDesc{};
        d.seg = -1; d.cand = int32_t(i); d.cnst = in_host[i].cnst;
        // at or beyond +-L N every shifted position falls outside the frame, and so does it for any larger offset: held to
        // that range, the kernel's index sums stay inside the int range whatever the caller's tag says
        const int32_t span = int32_t(std::min<int64_t>(int64_t(h->rx.cfg.frame_length) * h->rx.cfg.fft_len, 0x3fffffff));
        d.carr_offset = std::min(std::max(in_host[i].carr_offset, -span), span);
        d.n_syms = std::max(0, std::min(in_host[i].n_payload_syms, h->rx.frame_capacity()));
    }
    return h->rx.equalize_frames(reinterpret_cast<const dtl::cf32*>(d_frames), reinterpret_cast<const dtl::cf32*>(d_taps),
                                 descs.data(), int(n), reinterpret_cast<dtl::cf32*>(d_eq_hard),
                                 reinterpret_cast<dtl::cf32*>(d_eq_soft), reinterpret_cast<dtl::cf32*>(d_soft_syms),
                                 reinterpret_cast<dtl::FrameStat*>(out_host), static_cast<hipStream_t>(stream));
    DTL_API_END
}

int dtl_tb_gather(const int32_t* code_n, const int32_t* code_k, int32_t n_codes, int32_t frame_capacity, int32_t max_bps,
                  const dtl_rx_frame_info* frames_host, int64_t n_frames, dtl_cw_desc* d_cw_desc, int64_t cw_cap,
                  int64_t* n_cw, dtl_tb_span* spans_host, int64_t span_cap, int64_t* n_spans, void* stream)
{
    DTL_API_BEGIN("dtl_tb_gather")
    if (!code_n || !code_k || n_codes <= 0 || frame_capacity <= 0 || max_bps < 1 || max_bps > 4 || n_frames < 0 ||
        (n_frames > 0 && !frames_host) || !n_cw || !n_spans) {
        dtl::set_last_error("tb_gather: bad argument");
        return DTL_ERR_INVALID_ARG;
    }
    std::vector<std::pair<int, int>> nk;
    for (int i = 0; i < n_codes; i++) nk.push_back({ code_n[i], code_k[i] });
    std::vector<dtl::TbGather> gathers;
    std::vector<dtl::TbSpan> spans;
    int rc = dtl::tb_gather_frames(nk, frame_capacity, max_bps, reinterpret_cast<const dtl::FrameInfo*>(frames_host), n_frames,
                                   gathers, spans);
    if (rc != DTL_OK) return rc;
    int64_t total_cw = 0;
    for (auto& sp : spans) total_cw += sp.n_cw;
    *n_cw = total_cw;
    *n_spans = int64_t(spans.size());
    if (int64_t(spans.size()) > span_cap || total_cw > cw_cap) { dtl::set_last_error("tb_gather: output capacity too small"); return DTL_ERR_CAPACITY; }
    if (spans.empty()) return DTL_OK;
    if (!spans_host) { dtl::set_last_error("tb_gather: null output"); return DTL_ERR_INVALID_ARG; }
    std::memcpy(spans_host, spans.data(), spans.size() * sizeof(dtl::TbSpan));
    if (!d_cw_desc) return DTL_OK; // block bookkeeping only (host work: no device needed)
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device"); return DTL_ERR_NO_DEVICE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    dtl::DevBuf d_g;
    rc = d_g.ensure(gathers.size() * sizeof(dtl::TbGather));
    if (rc != DTL_OK) return rc;
    DTL_HIP_CHECK(hipMemcpyAsync(d_g.p, gathers.data(), gathers.size() * sizeof(dtl::TbGather), hipMemcpyHostToDevice, st));
    rc = dtl::launch_expand_cw(d_g.as<dtl::TbGather>(), int(gathers.size()), reinterpret_cast<dtl::CwDesc*>(d_cw_desc), st);
    if (rc != DTL_OK) return rc;
    DTL_HIP_CHECK(hipStreamSynchronize(st));
    return DTL_OK;
    DTL_API_END
}

int dtl_copy_device(void* dst, const void* src, size_t bytes, void* stream)
{
    DTL_API_BEGIN("dtl_copy_device")
    if (!bytes) return DTL_OK;
    if (!dst || !src) return DTL_ERR_INVALID_ARG;
    DTL_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return DTL_OK;
    DTL_API_END
}

int dtl_fix_sync(int32_t frame_len, const int64_t* triggers, int64_t n_triggers, int64_t stream_len,
                 int64_t* out, int64_t out_cap, int64_t* n_out)
{
    DTL_API_BEGIN("dtl_fix_sync")
    if (!n_out || (n_triggers && !triggers)) return DTL_ERR_INVALID_ARG;
    dtl::FrameDetectState st;
    st.frame_len = frame_len;
    std::vector<int64_t> res = dtl::fix_sync_sparse(st, std::vector<int64_t>(triggers, triggers + n_triggers), stream_len);
    *n_out = int64_t(res.size());
    if (int64_t(res.size()) > out_cap) return DTL_ERR_CAPACITY;
    std::copy(res.begin(), res.end(), out);
    return DTL_OK;
    DTL_API_END
}

int dtl_sync_pretest_check(const float* d_iq, int64_t n, int32_t fft_len, float threshold, uint64_t* counts_host, void* stream)
{
    DTL_API_BEGIN("dtl_sync_pretest_check")
    if (!counts_host || n < 0 || (n && !d_iq)) { dtl::set_last_error("sync_pretest_check: bad argument"); return DTL_ERR_INVALID_ARG; }
    if (dtl_device_count() <= 0) { dtl::set_last_error("no HIP device: the HIP path has no CPU fallback"); return DTL_ERR_NO_DEVICE; }
    counts_host[0] = counts_host[1] = counts_host[2] = 0;
    if (n == 0) return DTL_OK;
    if (n > int64_t(0x7fffffff)) { dtl::set_last_error("sync_pretest_check: at most 2^31-1 samples per call"); return DTL_ERR_INVALID_ARG; }
    auto* st = static_cast<hipStream_t>(stream);
    dtl::DevBuf d_edges, d_cnt, d_chk;
    const unsigned max_edges = unsigned(std::min<int64_t>(std::max<int64_t>(n / 2, 1024), 1 << 26));
    int rc = d_edges.ensure(size_t(max_edges) * 8);
    if (rc == DTL_OK) rc = d_cnt.ensure(16);
    if (rc == DTL_OK) rc = d_chk.ensure(32);
    if (rc != DTL_OK) return rc;
    DTL_HIP_CHECK(hipMemsetAsync(d_cnt.p, 0, 16, st));
    DTL_HIP_CHECK(hipMemsetAsync(d_chk.p, 0, 32, st));
    rc = dtl::launch_sync_edges(reinterpret_cast<const dtl::cf32*>(d_iq), n, 0, fft_len, threshold, d_edges.as<unsigned long long>(),
                                d_cnt.as<unsigned int>(), max_edges, nullptr, st, d_chk.as<unsigned long long>());
    if (rc != DTL_OK) return rc;
    unsigned long long c[3] = { 0, 0, 0 };
    DTL_HIP_CHECK(hipMemcpyAsync(c, d_chk.p, sizeof(c), hipMemcpyDeviceToHost, st));
    DTL_HIP_CHECK(hipStreamSynchronize(st));
    for (int i = 0; i < 3; i++) counts_host[i] = uint64_t(c[i]);
    return DTL_OK;
    DTL_API_END
}

int dtl_fix_sync_chunked(int32_t frame_len, const int64_t* triggers, int64_t n_triggers, const int64_t* buf_ends,
                         int64_t n_bufs, const int64_t* horizons, int64_t n_horizons, int64_t* out, int64_t out_cap,
                         int64_t* n_out)
{
    DTL_API_BEGIN("dtl_fix_sync_chunked")
    if (!n_out || (n_triggers && !triggers) || (n_bufs && !buf_ends) || (n_horizons && !horizons)) return DTL_ERR_INVALID_ARG;
    dtl::FrameDetectState st;
    st.frame_len = frame_len;
    std::vector<int64_t> res;
    int64_t begin = 0, ti = 0, hi = 0;
    for (int64_t b = 0; b < n_bufs; b++) {
        const int64_t end = buf_ends[b];
        if (end < begin) return DTL_ERR_INVALID_ARG;
        dtl::FixSyncRun run(st);
        // the triggers of this buffer, piece by piece: everything in front of the next horizon, then the horizon itself
        while (hi < n_horizons && horizons[hi] <= begin) hi++;
        for (; hi < n_horizons && horizons[hi] < end; hi++) {
            std::vector<int64_t> piece;
            while (ti < n_triggers && triggers[ti] < horizons[hi]) piece.push_back(triggers[ti++] - begin);
            run.feed(piece.data(), piece.size(), horizons[hi] - begin);
        }
        std::vector<int64_t> rest;
        while (ti < n_triggers && triggers[ti] < end) rest.push_back(triggers[ti++] - begin);
        run.feed(rest.data(), rest.size(), 0);
        for (int64_t p : run.finish(end - begin)) res.push_back(p + begin);
        begin = end;
    }
    *n_out = int64_t(res.size());
    if (int64_t(res.size()) > out_cap) return DTL_ERR_CAPACITY;
    std::copy(res.begin(), res.end(), out);
    return DTL_OK;
    DTL_API_END
}

int dtl_plateau_detect(const float* metric, int64_t n, float thr, int32_t max_len, int64_t* out, int64_t out_cap,
                       int64_t* n_out)
{
    DTL_API_BEGIN("dtl_plateau_detect")
    if (!metric || !n_out) return DTL_ERR_INVALID_ARG;
    std::vector<unsigned long long> edges;
    bool prev = false;
    for (int64_t i = 0; i < n; i++) {
        const bool cur = metric[i] >= thr;
        if (cur != prev) edges.push_back((static_cast<unsigned long long>(i) << 1) | (cur ? 1ull : 0ull));
        prev = cur;
    }
    std::vector<int64_t> res = dtl::plateau_from_edges(std::move(edges), n, max_len);
    *n_out = int64_t(res.size());
    if (int64_t(res.size()) > out_cap) return DTL_ERR_CAPACITY;
    std::copy(res.begin(), res.end(), out);
    return DTL_OK;
    DTL_API_END
}

int dtl_feedback_sequence(int32_t n_mcs, const double* mcs_snr, const int32_t* mcs_cnst, const int32_t* mcs_fec,
                          double hysteresis, int32_t decision_th, int32_t initial_mcs, const double* snr, int64_t n,
                          int32_t* out_cnst, int32_t* out_fec)
{
    DTL_API_BEGIN("dtl_feedback_sequence")
    if (n_mcs <= 0 || !mcs_snr || !mcs_cnst || !mcs_fec || !snr || !out_cnst || !out_fec) return DTL_ERR_INVALID_ARG;
    std::vector<dtl::McsEntry> lut;
    for (int i = 0; i < n_mcs; i++) lut.push_back({ mcs_snr[i], mcs_cnst[i], mcs_fec[i] });
    dtl::FeedbackDecision fb;
    fb.init(lut, hysteresis, decision_th, initial_mcs);
    for (int64_t i = 0; i < n; i++) fb.get(snr[i], &out_cnst[i], &out_fec[i]);
    return DTL_OK;
    DTL_API_END
}

} // extern "C"
