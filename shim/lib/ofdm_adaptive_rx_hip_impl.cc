/* gr::block over dtl_rx_process_stream.  Ports and message ports are those of the reference's hier block
 * (python/dtl/ofdm_adaptive_rx.py:42-59, python/dtl/ofdm_receiver.py:18-24, grc/dtl_ofdm_adaptive_rx.block.yml:22-53):
 *   in 0  gr_complex                         out 0  byte     decoded bytes (+ len tag per transport block)
 *   out 1 gr_complex  feedback TX (unused: the side-channel modem stays in the Python hier block)
 *   out 2 byte  trigger (frame_detect_bb)    out 3  byte     trigger (ofdm_sync_sc_cfb)
 *   out 4 gr_complex x fft_len  equaliser out0   out 5  gr_complex  equaliser out1, flattened
 *   out 6 float  fine frequency offset       msg: monitor, feedback, header
 *
 * Data path (options::async, the default).  A GNU Radio scheduler hands a block a few thousand samples per call; the GPU
 * wants millions.  general_work therefore only STAGES: it copies its samples into one of two page-locked slots and
 * issues the host-to-device copy at once, on a copy stream of its own.  A worker thread owned by the block takes
 * whatever a slot has accumulated when it becomes free -- small batches when the flowgraph idles, the whole slot under
 * load -- and runs ONE dtl_rx_process_stream call on it; meanwhile general_work fills the other slot, so the copy of
 * batch k+1 runs under the decode of batch k.  What a batch produced (bytes, equaliser rows, triggers, tags, messages)
 * is queued and leaves through the output buffers of later general_work calls, as much as each has room for.
 * The receiver's streaming call returns the same frames whatever the cuts (include/dtl_rx.h), so batch boundaries --
 * which depend on timing here -- do not show in the output.
 *
 * Tags.  Port 0: one packet-length tag per delivered transport block (FEC: fec_decoder_impl.cc:176; no FEC:
 * frame_pack_bb_impl.cc:118-121).  Ports 4 and 5 carry, on the first item of every frame, what
 * ofdm_adaptive_frame_equalizer_vcvc puts on its two outputs (frame_equalizer_vcvc_impl.cc:218-251): the header parser's
 * tags (packet_header.cc:202-229,281-312) handed down by header_payload_demux, "ofdm_sync_carr_offset", the noise estimate,
 * the tagged-stream length tag -- and, with options::propagate_feedback_tags, the SNR estimate and the feedback decision
 * (of the seven keys qa_ofdm_adaptive_frame_equalizer_vcvc.py:155-161 counts on a frame fed with constellation and payload
 * length only, six: the seventh, the channel state at the end of the frame, is not exported by the kernels and the
 * receive graph does not ask for it, python/dtl/ofdm_receiver.py:172). */
#include "ofdm_adaptive_rx_hip_impl.h"

#include <gnuradio/io_signature.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace gr {
namespace dtl_hip {

static const pmt::pmt_t MONITOR_PORT = pmt::mp("monitor");
static const pmt::pmt_t FEEDBACK_PORT = pmt::mp("feedback");
static const pmt::pmt_t HEADER_PORT = pmt::mp("header");
/* lib/dtl/ofdm_adaptive_utils.cc:23-39, lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:25-26 */
static const pmt::pmt_t CONSTELLATION_TAG_KEY = pmt::mp("constellation_tag_key");
static const pmt::pmt_t ESTIMATED_SNR_TAG_KEY = pmt::mp("estimated_snr_tag_key");
static const pmt::pmt_t NOISE_TAG_KEY = pmt::mp("noise_tag_key");
static const pmt::pmt_t FEEDBACK_CONSTELLATION_KEY = pmt::mp("feedback_constellation_key");
static const pmt::pmt_t PAYLOAD_LENGTH_KEY = pmt::mp("payload_length_key");
static const pmt::pmt_t FEC_KEY = pmt::mp("fec_key");
static const pmt::pmt_t FEC_FEEDBACK_KEY = pmt::mp("fec_feedback_key");
static const pmt::pmt_t FEC_TB_KEY = pmt::mp("fec_tb_key");
static const pmt::pmt_t FEC_OFFSET_KEY = pmt::mp("fec_offset_key");
static const pmt::pmt_t FEC_TB_PAYLOAD_KEY = pmt::mp("fec_tb_payload_key");
static const pmt::pmt_t CARR_OFFSET_KEY = pmt::mp("ofdm_sync_carr_offset");

#define DTL_HIP_CALL(e)                                                             \
    do {                                                                            \
        if ((e) != DTL_OK) throw std::runtime_error(dtl_last_error()); /* reference: exceptions out of work() */ \
    } while (0)
#define HIP_CALL(e)                                                                 \
    do {                                                                            \
        hipError_t _e = (e);                                                        \
        if (_e != hipSuccess) throw std::runtime_error(std::string("ofdm_adaptive_rx_hip: " #e ": ") + hipGetErrorString(_e)); \
    } while (0)

ofdm_adaptive_rx_hip::sptr ofdm_adaptive_rx_hip::make(const dtl_rx_cfg& cfg, const std::string& len_tag_key)
{
    options opt;
    opt.len_tag_key = len_tag_key;
    return gnuradio::make_block_sptr<ofdm_adaptive_rx_hip_impl>(cfg, opt);
}

ofdm_adaptive_rx_hip::sptr ofdm_adaptive_rx_hip::make(const dtl_rx_cfg& cfg, const options& opt)
{
    return gnuradio::make_block_sptr<ofdm_adaptive_rx_hip_impl>(cfg, opt);
}

ofdm_adaptive_rx_hip_impl::ofdm_adaptive_rx_hip_impl(const dtl_rx_cfg& cfg, const options& opt)
    : gr::block("ofdm_adaptive_rx_hip", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                gr::io_signature::makev(7, 7,
                                        std::vector<int>{ int(sizeof(char)), int(sizeof(gr_complex)), int(sizeof(char)), int(sizeof(char)),
                                                          int(sizeof(gr_complex)) * cfg.fft_len, int(sizeof(gr_complex)), int(sizeof(float)) })),
      d_opt(opt), d_fft_len(cfg.fft_len), d_frame_length(cfg.frame_length),
      d_frame_samples((2 + (cfg.has_fec ? 2 : 1) + cfg.frame_length) * (cfg.fft_len + cfg.cp_len)), d_has_fec(cfg.has_fec),
      d_len_key(pmt::string_to_symbol(opt.len_tag_key)), d_frame_len_key(pmt::string_to_symbol(opt.frame_length_tag_key)),
      d_frame_no_key(pmt::string_to_symbol(opt.frame_no_tag_key))
{
    /* ofdm_adaptive_frame_equalizer_vcvc's first switch re-publishes the channel state at the END of a frame
     * (frame_equalizer_vcvc_impl.cc:186-193,231-236); the receive graph passes False (python/dtl/ofdm_receiver.py:172) and
     * the kernels do not export that state */
    if (d_opt.propagate_channel_state) throw std::invalid_argument("ofdm_adaptive_rx_hip: propagate_channel_state is not supported");
    if (d_opt.max_batch_samples < 4 * int64_t(d_frame_samples)) d_opt.max_batch_samples = 4 * int64_t(d_frame_samples);
    if (d_opt.max_batch_samples > (int64_t(1) << 30)) d_opt.max_batch_samples = int64_t(1) << 30;
    DTL_HIP_CALL(dtl_rx_create(&cfg, &d_rx));
    if (cfg.device >= 0) HIP_CALL(hipSetDevice(cfg.device));
    HIP_CALL(hipStreamCreateWithFlags(&d_stream, hipStreamNonBlocking));
    HIP_CALL(hipStreamCreateWithFlags(&d_copy, hipStreamNonBlocking));
    HIP_CALL(hipEventCreateWithFlags(&d_copied, hipEventDisableTiming));
    for (auto& s : d_slot) {
        HIP_CALL(hipHostMalloc(reinterpret_cast<void**>(&s.pinned), size_t(d_opt.max_batch_samples) * sizeof(gr_complex), hipHostMallocDefault));
        HIP_CALL(hipMalloc(reinterpret_cast<void**>(&s.dev), size_t(d_opt.max_batch_samples) * sizeof(gr_complex)));
    }
    set_tag_propagation_policy(TPP_DONT);
    message_port_register_out(MONITOR_PORT);
    message_port_register_out(FEEDBACK_PORT);
    message_port_register_out(HEADER_PORT);
    set_output_multiple(1);
    /* the GPU wants large buffers: ask the scheduler for at least eight frames per call */
    set_min_noutput_items(8 * d_frame_samples);
    if (d_opt.async) {
        int dev = 0;
        HIP_CALL(hipGetDevice(&dev));
        d_worker = std::thread([this, dev] { (void)hipSetDevice(dev); worker_loop(); });
    }
}

ofdm_adaptive_rx_hip_impl::~ofdm_adaptive_rx_hip_impl()
{
    if (d_worker.joinable()) {
        { std::lock_guard<std::mutex> lk(d_mu); d_quit = true; }
        d_cv_work.notify_all();
        d_worker.join();
    }
    dtl_rx_destroy(d_rx);
    for (auto& s : d_slot) {
        if (s.dev) (void)hipFree(s.dev);
        if (s.pinned) (void)hipHostFree(s.pinned);
    }
    if (d_out_pinned) (void)hipHostFree(d_out_pinned);
    if (d_copied) (void)hipEventDestroy(d_copied);
    if (d_copy) (void)hipStreamDestroy(d_copy);
    if (d_stream) (void)hipStreamDestroy(d_stream);
}

void ofdm_adaptive_rx_hip_impl::forecast(int /*noutput_items*/, gr_vector_int& ninput_items_required)
{
    /* output of earlier batches may be waiting: then the block wants to be called even without new input */
    const bool pending = !d_pending_bytes.empty() || !d_pending_hard.empty() || !d_pending_soft.empty() || d_emitted < d_decoded_end - d_frame_samples;
    ninput_items_required[0] = pending ? 0 : 1;
}

void ofdm_adaptive_rx_hip_impl::ensure_out_pinned(size_t bytes)
{
    if (bytes <= d_out_pinned_cap) return;
    if (d_out_pinned) (void)hipHostFree(d_out_pinned);
    d_out_pinned = nullptr; d_out_pinned_cap = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    HIP_CALL(hipHostMalloc(&d_out_pinned, want, hipHostMallocDefault));
    d_out_pinned_cap = want;
}

/* under d_mu: hand the filled slot to the decoder, staging goes on in the other one */
bool ofdm_adaptive_rx_hip_impl::take_batch(int& b, int64_t& n)
{
    b = d_fill;
    n = d_slot[b].fill;
    if (n == 0) return false;
    d_fill ^= 1;
    /* behind every copy general_work has issued for this slot (they were issued under d_mu) */
    HIP_CALL(hipEventRecord(d_copied, d_copy));
    return true;
}

/* one dtl_rx_process_stream call over slot b (b < 0: the final flush) and everything it produced, copied out of the
 * handle's buffers (they are valid until the next call on the handle) */
void ofdm_adaptive_rx_hip_impl::run_batch(int b, int64_t n, bool final)
{
    dtl_rx_result r;
    const float* d_in = nullptr;
    if (b >= 0) {
        HIP_CALL(hipStreamWaitEvent(d_stream, d_copied, 0));
        d_in = reinterpret_cast<const float*>(d_slot[b].dev);
    }
    DTL_HIP_CALL(dtl_rx_process_stream(d_rx, d_in, n, DTL_RX_WANT_EQ_HARD | DTL_RX_WANT_EQ_SOFT, final ? 1 : 0, d_stream, &r));
    batch_out o;
    d_pos += n;
    o.end = d_pos;
    o.final = final;
    const size_t n_bytes = size_t(r.n_payload_bytes);
    const size_t row = size_t(d_frame_length) * size_t(d_fft_len);
    const size_t n_rows = size_t(r.n_payload_frames) * row;
    /* device -> pinned host in one go on the decode stream, one synchronisation */
    const size_t o_hard = (n_bytes + 63) & ~size_t(63), o_soft = o_hard + n_rows * sizeof(gr_complex);
    ensure_out_pinned(o_soft + n_rows * sizeof(gr_complex) + 64);
    auto* base = static_cast<unsigned char*>(d_out_pinned);
    if (n_bytes) HIP_CALL(hipMemcpyAsync(base, r.d_payload, n_bytes, hipMemcpyDeviceToHost, d_stream));
    if (n_rows) {
        HIP_CALL(hipMemcpyAsync(base + o_hard, r.d_eq_hard, n_rows * sizeof(gr_complex), hipMemcpyDeviceToHost, d_stream));
        HIP_CALL(hipMemcpyAsync(base + o_soft, r.d_eq_soft, n_rows * sizeof(gr_complex), hipMemcpyDeviceToHost, d_stream));
    }
    /* the host records meanwhile */
    o.tbs.assign(r.tbs, r.tbs + r.n_tbs);
    o.frames.resize(size_t(r.n_frames));
    for (int64_t f = 0; f < r.n_frames; f++) o.frames[size_t(f)].fi = r.frames[f];
    o.raw.assign(r.raw_triggers, r.raw_triggers + r.n_raw_triggers);
    o.fixed.assign(r.fixed_triggers, r.fixed_triggers + r.n_fixed_triggers);
    o.angle.assign(r.trigger_angles, r.trigger_angles + r.n_raw_triggers);
    HIP_CALL(hipStreamSynchronize(d_stream));
    o.bytes.assign(base, base + n_bytes);
    const auto* hard = reinterpret_cast<const gr_complex*>(base + o_hard);
    const auto* soft = reinterpret_cast<const gr_complex*>(base + o_soft);
    o.hard.assign(hard, hard + n_rows);
    o.soft.assign(soft, soft + n_rows);
    {
        std::lock_guard<std::mutex> lk(d_mu);
        if (b >= 0) d_slot[b].fill = 0;
        d_results.push_back(std::move(o));
    }
    d_cv_space.notify_all();
}

void ofdm_adaptive_rx_hip_impl::worker_loop()
{
    for (;;) {
        int b = -1;
        int64_t n = 0;
        bool have = false, final = false;
        {
            std::unique_lock<std::mutex> lk(d_mu);
            d_cv_work.wait(lk, [&] { return d_quit || d_flush || d_slot[d_fill].fill > 0; });
            try {
                have = take_batch(b, n);
            } catch (const std::exception& e) {
                d_error = e.what();
                d_slot[d_fill].fill = 0;
            }
            if (!have) {
                if (d_flush) { final = true; d_flush = false; }
                else if (d_quit) return;
                else continue;
            }
            d_busy = true;
        }
        try {
            if (have) run_batch(b, n, false);
            else if (final) run_batch(-1, 0, true);
        } catch (const std::exception& e) {
            std::lock_guard<std::mutex> lk(d_mu);
            d_error = e.what();
            if (b >= 0) d_slot[b].fill = 0;
        }
        {
            std::lock_guard<std::mutex> lk(d_mu);
            d_busy = false;
        }
        d_cv_idle.notify_all();
        d_cv_space.notify_all();
    }
}

void ofdm_adaptive_rx_hip_impl::wait_idle()
{
    if (!d_opt.async) return;
    std::unique_lock<std::mutex> lk(d_mu);
    d_cv_work.notify_all();
    d_cv_idle.wait(lk, [&] { return !d_busy && !d_flush && d_slot[d_fill].fill == 0; });
}

void ofdm_adaptive_rx_hip_impl::queue_tag(int port, uint64_t offset, const pmt::pmt_t& key, const pmt::pmt_t& value)
{
    tag_t t;
    t.offset = offset; t.key = key; t.value = value; t.srcid = pmt::string_to_symbol(name());
    d_pending_tags[port].push_back(t);
}

/* tags of the items [.., upto) of `port` go to the runtime now (their items are being produced by this call) */
int ofdm_adaptive_rx_hip_impl::emit_tags(int port, uint64_t upto)
{
    int n = 0;
    auto& q = d_pending_tags[port];
    while (!q.empty() && q.front().offset < upto) {
        add_item_tag(unsigned(port), q.front());
        q.pop_front();
        n++;
    }
    return n;
}

/* messages, tags and port data of one decoded batch -> the block's queues (general_work's thread) */
void ofdm_adaptive_rx_hip_impl::queue_batch(const batch_out& r)
{
    const uint64_t row_items = uint64_t(d_frame_length), row_samples = uint64_t(d_frame_length) * uint64_t(d_fft_len);
    for (const auto& fo : r.frames) {
        const dtl_rx_frame_info& fi = fo.fi;
        if (!fi.header_ok) { message_port_pub(HEADER_PORT, pmt::get_PMT_F()); continue; }  /* packet_headerparser_b: PMT_F, the demux skips */
        /* the header parser's tags (packet_header.cc:202-229, 281-312), as the "header_data" dictionary and -- handed down
         * by the demux and the frame equaliser -- as tags on both equaliser outputs */
        std::vector<std::pair<pmt::pmt_t, pmt::pmt_t>> hdr;
        if (d_has_fec) {
            hdr.emplace_back(FEC_TB_KEY, pmt::from_long(fi.tb_no));
            hdr.emplace_back(FEC_FEEDBACK_KEY, pmt::from_long(fi.fec_fb));
            hdr.emplace_back(FEC_OFFSET_KEY, pmt::from_long(fi.tb_offset));
            hdr.emplace_back(FEC_KEY, pmt::from_long(fi.fec));
            hdr.emplace_back(FEC_TB_PAYLOAD_KEY, pmt::from_long(fi.tb_payload));
        }
        hdr.emplace_back(PAYLOAD_LENGTH_KEY, pmt::from_long(fi.payload_len));
        hdr.emplace_back(d_len_key, pmt::from_long(fi.n_payload_syms));
        hdr.emplace_back(d_frame_no_key, pmt::from_long(fi.frame_no));
        hdr.emplace_back(CONSTELLATION_TAG_KEY, pmt::from_long(fi.cnst));
        hdr.emplace_back(d_frame_len_key, pmt::from_long(d_frame_length));
        hdr.emplace_back(FEEDBACK_CONSTELLATION_KEY, pmt::from_long(fi.fb_cnst));
        hdr.emplace_back(CARR_OFFSET_KEY, pmt::from_long(fi.carr_offset));           /* ofdm_chanest_vcvc's tags ride along */
        pmt::pmt_t h = pmt::make_dict();
        for (auto& kv : hdr) h = pmt::dict_add(h, kv.first, kv.second);
        message_port_pub(HEADER_PORT, h);
        if (fi.payload_index < 0) continue;
        /* frame equaliser: feedback_port (frame_equalizer_vcvc_impl.cc:196-208) and monitor (:210-216) */
        pmt::pmt_t fb = pmt::make_dict();
        fb = pmt::dict_add(fb, FEEDBACK_CONSTELLATION_KEY, pmt::from_long(static_cast<unsigned char>(fi.feedback_cnst)));
        fb = pmt::dict_add(fb, FEC_FEEDBACK_KEY, pmt::from_long(fi.feedback_fec));
        message_port_pub(FEEDBACK_PORT, fb);
        d_frames_count += (fi.lost_frames - d_lost_total) + 1;                    /* :124-138: lost frames + this one */
        d_lost_total = fi.lost_frames;
        pmt::pmt_t m = pmt::make_dict();
        m = pmt::dict_add(m, pmt::mp("constellation_key"), pmt::from_long(static_cast<unsigned char>(fi.feedback_cnst)));
        m = pmt::dict_add(m, FEC_KEY, pmt::from_long(fi.feedback_fec));
        m = pmt::dict_add(m, ESTIMATED_SNR_TAG_KEY, pmt::from_double(fi.snr_db));
        m = pmt::dict_add(m, NOISE_TAG_KEY, pmt::from_double(fi.noise));
        m = pmt::dict_add(m, pmt::mp("lost_frames_rate"), pmt::from_double(100.0 * double(d_lost_total) / double(d_frames_count)));
        message_port_pub(MONITOR_PORT, m);
        if (fi.payload_len == 0) continue;                                          /* work() returns 0 items: no tags (:218,254) */
        /* tags on the frame's first item of both equaliser outputs (:218-251): every input tag but the channel state and
         * the tagged-stream length tag, the noise estimate, [the feedback tags,] and the length tag the tagged-stream
         * block base puts on each output */
        const uint64_t off[2] = { d_queued[4] + uint64_t(fi.payload_index) * row_items, d_queued[5] + uint64_t(fi.payload_index) * row_samples };
        const int port[2] = { 4, 5 };
        for (int oi = 0; oi < 2; oi++) {
            for (auto& kv : hdr)
                if (kv.first != d_frame_len_key) queue_tag(port[oi], off[oi], kv.first, kv.second);
            queue_tag(port[oi], off[oi], NOISE_TAG_KEY, pmt::from_double(fi.noise));
            if (d_opt.propagate_feedback_tags) {
                queue_tag(port[oi], off[oi], ESTIMATED_SNR_TAG_KEY, pmt::from_double(fi.snr_db));
                queue_tag(port[oi], off[oi], FEEDBACK_CONSTELLATION_KEY, pmt::from_long(static_cast<unsigned char>(fi.feedback_cnst)));
                queue_tag(port[oi], off[oi], FEC_FEEDBACK_KEY, pmt::from_long(fi.feedback_fec));
            }
            queue_tag(port[oi], off[oi], d_frame_len_key, pmt::from_long(d_frame_length));
        }
    }
    /* ---- port 0: decoded bytes + one length tag per delivered transport block; FEC decoder monitor message
     * (fec_decoder_impl.cc:183-195) */
    for (const auto& tb : r.tbs) {
        if ((d_has_fec ? tb.crc_ok != 0 : true) && tb.out_offset >= 0 && tb.user_bits >= 8)
            queue_tag(0, d_queued[0] + uint64_t(tb.out_offset), d_len_key, pmt::from_long(tb.user_bits / 8));
        pmt::pmt_t m = pmt::make_dict();
        m = pmt::dict_add(m, pmt::mp("tb_no"), pmt::from_long(tb.tb_no));
        m = pmt::dict_add(m, pmt::mp("tb_payload"), pmt::from_long(tb.tb_payload));
        m = pmt::dict_add(m, pmt::mp("crc_ok"), pmt::from_bool(tb.crc_ok != 0));
        m = pmt::dict_add(m, pmt::mp("avg_it"), pmt::from_long(tb.avg_it));
        message_port_pub(MONITOR_PORT, m);
    }
    d_pending_bytes.insert(d_pending_bytes.end(), r.bytes.begin(), r.bytes.end());
    d_queued[0] += r.bytes.size();
    /* ---- ports 4, 5 */
    d_pending_hard.insert(d_pending_hard.end(), r.hard.begin(), r.hard.end());
    d_pending_soft.insert(d_pending_soft.end(), r.soft.begin(), r.soft.end());
    d_queued[4] += r.hard.size() / uint64_t(d_fft_len);
    d_queued[5] += r.soft.size();
    /* ---- ports 2, 3, 6 */
    for (size_t i = 0; i < r.raw.size(); i++) { d_raw.push_back(r.raw[i]); d_angle.push_back(r.angle[i]); }
    for (int64_t t : r.fixed) d_fixed.push_back(t);
    d_decoded_end = r.end;
    if (r.final) d_final_seen = true;
}

void ofdm_adaptive_rx_hip_impl::drain_results()
{
    std::deque<batch_out> got;
    std::string err;
    {
        std::lock_guard<std::mutex> lk(d_mu);
        got.swap(d_results);
        err.swap(d_error);
    }
    for (const auto& r : got) queue_batch(r);
    if (!err.empty()) throw std::runtime_error(err);
}

int ofdm_adaptive_rx_hip_impl::general_work(int noutput_items, gr_vector_int& ninput_items,
                                            gr_vector_const_void_star& input_items, gr_vector_void_star& output_items)
{
    const int64_t n = ninput_items[0];
    int64_t taken = 0;
    if (n > 0) {
        std::unique_lock<std::mutex> lk(d_mu);
        const int64_t cap = d_opt.max_batch_samples;
        if (d_opt.async)   /* both slots full: wait for the decoder to take one (back-pressure on the flowgraph) */
            d_cv_space.wait(lk, [&] { return d_slot[d_fill].fill < cap || !d_error.empty(); });
        slot& s = d_slot[d_fill];
        taken = std::min<int64_t>(n, cap - s.fill);
        if (taken > 0) {
            std::memcpy(s.pinned + s.fill, input_items[0], size_t(taken) * sizeof(gr_complex));
            HIP_CALL(hipMemcpyAsync(s.dev + s.fill, s.pinned + s.fill, size_t(taken) * sizeof(gr_complex), hipMemcpyHostToDevice, d_copy));
            s.fill += taken;
        }
        if (!d_opt.async) {
            int b;
            int64_t nb;
            if (take_batch(b, nb)) { lk.unlock(); run_batch(b, nb, false); }
        }
    }
    if (d_opt.async) d_cv_work.notify_one();
    drain_results();
    /* ---- hand over what fits */
    const size_t room = size_t(noutput_items);
    const size_t nb = std::min(d_pending_bytes.size(), room);
    std::copy(d_pending_bytes.begin(), d_pending_bytes.begin() + long(nb), static_cast<unsigned char*>(output_items[0]));
    d_pending_bytes.erase(d_pending_bytes.begin(), d_pending_bytes.begin() + long(nb));
    emit_tags(0, nitems_written(0) + nb);
    const size_t nv = std::min(d_pending_hard.size() / size_t(d_fft_len), room);
    std::copy(d_pending_hard.begin(), d_pending_hard.begin() + long(nv * size_t(d_fft_len)), static_cast<gr_complex*>(output_items[4]));
    d_pending_hard.erase(d_pending_hard.begin(), d_pending_hard.begin() + long(nv * size_t(d_fft_len)));
    emit_tags(4, nitems_written(4) + nv);
    const size_t ns = std::min(d_pending_soft.size(), room);
    std::copy(d_pending_soft.begin(), d_pending_soft.begin() + long(ns), static_cast<gr_complex*>(output_items[5]));
    d_pending_soft.erase(d_pending_soft.begin(), d_pending_soft.begin() + long(ns));
    emit_tags(5, nitems_written(5) + ns);
    /* ---- ports 2, 3, 6: per-sample streams up to one frame behind the decoded input (everything after the flush) */
    const int64_t hi = d_final_seen ? d_decoded_end : std::max(d_emitted, d_decoded_end - d_frame_samples);
    const int64_t n_stream = std::min<int64_t>(hi - d_emitted, noutput_items);
    auto* fixed_out = static_cast<unsigned char*>(output_items[2]);
    auto* raw_out = static_cast<unsigned char*>(output_items[3]);
    auto* freq_out = static_cast<float*>(output_items[6]);
    std::memset(fixed_out, 0, size_t(n_stream));
    std::memset(raw_out, 0, size_t(n_stream));
    while (!d_fixed.empty() && d_fixed.front() < d_emitted + n_stream) {
        if (d_fixed.front() >= d_emitted) fixed_out[d_fixed.front() - d_emitted] = 1;
        d_fixed.pop_front();
    }
    for (int64_t i = 0; i < n_stream; i++) {
        while (!d_raw.empty() && d_raw.front() < d_emitted + i) { d_held = d_angle.front(); d_raw.pop_front(); d_angle.pop_front(); }  /* (cannot happen: kept in step) */
        if (!d_raw.empty() && d_raw.front() == d_emitted + i) { raw_out[i] = 1; d_held = d_angle.front(); d_raw.pop_front(); d_angle.pop_front(); }
        freq_out[i] = d_held;   /* sample_and_hold_ff */
    }
    d_emitted += n_stream;
    consume_each(int(taken));
    produce(0, int(nb)); produce(1, 0); produce(2, int(n_stream)); produce(3, int(n_stream));
    produce(4, int(nv)); produce(5, int(ns)); produce(6, int(n_stream));
    return WORK_CALLED_PRODUCE;
}

bool ofdm_adaptive_rx_hip_impl::stop()
{
    /* flush: the detector closes an open plateau, incomplete frames are dropped; what the flush decodes is queued like
     * any batch (a flowgraph that has stopped never asks for it; tests do, with one more general_work call) */
    if (d_opt.async) {
        wait_idle();
        { std::lock_guard<std::mutex> lk(d_mu); d_flush = true; }
        d_cv_work.notify_all();
        wait_idle();
    } else {
        run_batch(-1, 0, true);
    }
    return true;
}

} // namespace dtl_hip
} // namespace gr
