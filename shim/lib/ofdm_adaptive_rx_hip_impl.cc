/* gr::block forwarding every scheduler buffer to dtl_rx_process_stream.  Ports and message ports are those of the
 * reference's hier block (python/dtl/ofdm_adaptive_rx.py:42-59, python/dtl/ofdm_receiver.py:18-24):
 *   in 0  gr_complex                         out 0  byte     decoded bytes (+ len tag per transport block)
 *   out 1 gr_complex  feedback TX (unused: the side-channel modem stays in the Python hier block)
 *   out 2 byte  trigger (frame_detect_bb)    out 3  byte     trigger (ofdm_sync_sc_cfb)
 *   out 4 gr_complex x fft_len  equaliser out0   out 5  gr_complex  equaliser out1, flattened
 *   out 6 float  fine frequency offset       msg: monitor, feedback, header
 * NOT compiled in this repository (GNU Radio is not installed here); see shim/README.md. */
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

#define DTL_HIP_CALL(e)                                                             \
    do {                                                                            \
        if ((e) != DTL_OK) throw std::runtime_error(dtl_last_error()); /* reference: exceptions out of work() */ \
    } while (0)

ofdm_adaptive_rx_hip::sptr ofdm_adaptive_rx_hip::make(const dtl_rx_cfg& cfg, const std::string& len_tag_key)
{
    return gnuradio::make_block_sptr<ofdm_adaptive_rx_hip_impl>(cfg, len_tag_key);
}

ofdm_adaptive_rx_hip_impl::ofdm_adaptive_rx_hip_impl(const dtl_rx_cfg& cfg, const std::string& len_tag_key)
    : gr::block("ofdm_adaptive_rx_hip", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                gr::io_signature::makev(7, 7,
                                        std::vector<int>{ int(sizeof(char)), int(sizeof(gr_complex)), int(sizeof(char)), int(sizeof(char)),
                                                          int(sizeof(gr_complex)) * cfg.fft_len, int(sizeof(gr_complex)), int(sizeof(float)) })),
      d_fft_len(cfg.fft_len), d_frame_length(cfg.frame_length),
      d_frame_samples((2 + (cfg.has_fec ? 2 : 1) + cfg.frame_length) * (cfg.fft_len + cfg.cp_len)),
      d_len_key(pmt::string_to_symbol(len_tag_key))
{
    DTL_HIP_CALL(dtl_rx_create(&cfg, &d_rx));
    if (hipStreamCreate(&d_stream) != hipSuccess) throw std::runtime_error("hipStreamCreate");
    set_tag_propagation_policy(TPP_DONT);
    message_port_register_out(MONITOR_PORT);
    message_port_register_out(FEEDBACK_PORT);
    message_port_register_out(HEADER_PORT);
    /* the GPU wants large buffers: ask the scheduler for at least eight frames per call */
    set_output_multiple(1);
    set_min_noutput_items(8 * d_frame_samples);
}

ofdm_adaptive_rx_hip_impl::~ofdm_adaptive_rx_hip_impl()
{
    dtl_rx_destroy(d_rx);
    if (d_iq) (void)hipFree(d_iq);
    if (d_pinned) (void)hipHostFree(d_pinned);
    if (d_stream) (void)hipStreamDestroy(d_stream);
}

void ofdm_adaptive_rx_hip_impl::forecast(int noutput_items, gr_vector_int& ninput_items_required)
{
    ninput_items_required[0] = std::max(noutput_items, 1);
}

void ofdm_adaptive_rx_hip_impl::ensure(size_t n)
{
    if (n <= d_iq_cap) return;
    if (d_iq) (void)hipFree(d_iq);
    if (d_pinned) (void)hipHostFree(d_pinned);
    if (hipMalloc(&d_iq, n * sizeof(gr_complex)) != hipSuccess || hipHostMalloc(&d_pinned, n * sizeof(gr_complex)) != hipSuccess)
        throw std::runtime_error("ofdm_adaptive_rx_hip: out of device / pinned memory");
    d_iq_cap = n;
}

void ofdm_adaptive_rx_hip_impl::publish_messages(const dtl_rx_result& r)
{
    for (int64_t f = 0; f < r.n_frames; f++) {
        const dtl_rx_frame_info& fi = r.frames[f];
        if (!fi.header_ok) continue;
        /* packet_headerparser_b "header_data" (ofdm_adaptive_packet_header.cc:281-312) */
        pmt::pmt_t h = pmt::make_dict();
        h = pmt::dict_add(h, pmt::mp("frame_no"), pmt::from_long(fi.frame_no));
        h = pmt::dict_add(h, pmt::mp("payload_len"), pmt::from_long(fi.payload_len));
        h = pmt::dict_add(h, pmt::mp("constellation"), pmt::from_long(fi.cnst));
        message_port_pub(HEADER_PORT, h);
        if (fi.payload_index < 0) continue;
        /* frame equaliser: feedback_port (frame_equalizer_vcvc_impl.cc:218-231) and monitor (:196-216) */
        pmt::pmt_t fb = pmt::make_dict();
        fb = pmt::dict_add(fb, pmt::mp("feedback_constellation_key"), pmt::from_long(fi.feedback_cnst));
        fb = pmt::dict_add(fb, pmt::mp("fec_feedback_key"), pmt::from_long(fi.feedback_fec));
        message_port_pub(FEEDBACK_PORT, fb);
        pmt::pmt_t m = pmt::make_dict();
        m = pmt::dict_add(m, pmt::mp("estimated_snr_tag_key"), pmt::from_double(fi.snr_db));
        m = pmt::dict_add(m, pmt::mp("noise_tag_key"), pmt::from_double(fi.noise));
        m = pmt::dict_add(m, pmt::mp("lost_frames"), pmt::from_long(fi.lost_frames));
        message_port_pub(MONITOR_PORT, m);
    }
    for (int64_t t = 0; t < r.n_tbs; t++) { /* FEC decoder monitor message (fec_decoder_impl.cc:183-195) */
        pmt::pmt_t m = pmt::make_dict();
        m = pmt::dict_add(m, pmt::mp("tb_no"), pmt::from_long(r.tbs[t].tb_no));
        m = pmt::dict_add(m, pmt::mp("crc_ok"), pmt::from_bool(r.tbs[t].crc_ok != 0));
        m = pmt::dict_add(m, pmt::mp("avg_it"), pmt::from_long(r.tbs[t].avg_it));
        message_port_pub(MONITOR_PORT, m);
    }
}

int ofdm_adaptive_rx_hip_impl::general_work(int noutput_items, gr_vector_int& ninput_items,
                                            gr_vector_const_void_star& input_items, gr_vector_void_star& output_items)
{
    const int64_t n = ninput_items[0];
    if (n <= 0) return 0;
    ensure(size_t(n));
    std::memcpy(d_pinned, input_items[0], size_t(n) * sizeof(gr_complex));
    if (hipMemcpyAsync(d_iq, d_pinned, size_t(n) * sizeof(gr_complex), hipMemcpyHostToDevice, d_stream) != hipSuccess)
        throw std::runtime_error("ofdm_adaptive_rx_hip: host to device copy failed");
    dtl_rx_result r;
    const int64_t end = dtl_rx_stream_position(d_rx) + n;
    DTL_HIP_CALL(dtl_rx_process_stream(d_rx, static_cast<const float*>(d_iq), n, DTL_RX_WANT_EQ_HARD | DTL_RX_WANT_EQ_SOFT,
                                       /*final=*/0, d_stream, &r));
    /* ---- port 0: decoded bytes + one length tag per CRC-ok transport block */
    const size_t old = d_pending_bytes.size();
    d_pending_bytes.resize(old + size_t(r.n_payload_bytes));
    if (r.n_payload_bytes)
        (void)hipMemcpy(d_pending_bytes.data() + old, r.d_payload, size_t(r.n_payload_bytes), hipMemcpyDeviceToHost);
    for (int64_t t = 0; t < r.n_tbs; t++)
        if (r.tbs[t].crc_ok)
            add_item_tag(0, nitems_written(0) + old + uint64_t(r.tbs[t].out_offset), d_len_key, pmt::from_long(r.tbs[t].user_bits / 8));
    /* ---- ports 4, 5: equaliser outputs of the frames decoded by this call */
    const size_t vecs = size_t(r.n_payload_frames) * size_t(d_frame_length);
    const size_t h0 = d_pending_hard.size(), s0 = d_pending_soft.size();
    d_pending_hard.resize(h0 + vecs * size_t(d_fft_len));
    d_pending_soft.resize(s0 + vecs * size_t(d_fft_len));
    if (vecs) {
        (void)hipMemcpy(d_pending_hard.data() + h0, r.d_eq_hard, vecs * size_t(d_fft_len) * sizeof(gr_complex), hipMemcpyDeviceToHost);
        (void)hipMemcpy(d_pending_soft.data() + s0, r.d_eq_soft, vecs * size_t(d_fft_len) * sizeof(gr_complex), hipMemcpyDeviceToHost);
    }
    publish_messages(r);
    /* ---- ports 2, 3, 6: per-sample streams up to one frame behind the input */
    for (int64_t i = 0; i < r.n_raw_triggers; i++) { d_raw.push_back(r.raw_triggers[i]); d_angle.push_back(r.trigger_angles[i]); }
    for (int64_t i = 0; i < r.n_fixed_triggers; i++) d_fixed.push_back(r.fixed_triggers[i]);
    const int64_t hi = std::max(d_emitted, end - d_frame_samples);
    const int64_t n_stream = std::min<int64_t>(hi - d_emitted, noutput_items);
    auto* fixed_out = static_cast<unsigned char*>(output_items[2]);
    auto* raw_out = static_cast<unsigned char*>(output_items[3]);
    auto* freq_out = static_cast<float*>(output_items[6]);
    std::memset(fixed_out, 0, size_t(n_stream));
    std::memset(raw_out, 0, size_t(n_stream));
    while (!d_fixed.empty() && d_fixed.front() < d_emitted + n_stream) { fixed_out[d_fixed.front() - d_emitted] = 1; d_fixed.pop_front(); }
    for (int64_t i = 0; i < n_stream; i++) {
        if (!d_raw.empty() && d_raw.front() == d_emitted + i) { raw_out[i] = 1; d_held = d_angle.front(); d_raw.pop_front(); d_angle.pop_front(); }
        freq_out[i] = d_held;   /* sample_and_hold_ff */
    }
    d_emitted += n_stream;
    /* ---- hand over what fits */
    const size_t nb = std::min<size_t>(d_pending_bytes.size(), size_t(noutput_items));
    std::memcpy(output_items[0], d_pending_bytes.data(), nb);
    d_pending_bytes.erase(d_pending_bytes.begin(), d_pending_bytes.begin() + long(nb));
    const size_t nv = std::min<size_t>(d_pending_hard.size() / size_t(d_fft_len), size_t(noutput_items));
    std::memcpy(output_items[4], d_pending_hard.data(), nv * size_t(d_fft_len) * sizeof(gr_complex));
    d_pending_hard.erase(d_pending_hard.begin(), d_pending_hard.begin() + long(nv * size_t(d_fft_len)));
    const size_t ns = std::min<size_t>(d_pending_soft.size(), size_t(noutput_items));
    std::memcpy(output_items[5], d_pending_soft.data(), ns * sizeof(gr_complex));
    d_pending_soft.erase(d_pending_soft.begin(), d_pending_soft.begin() + long(ns));
    consume_each(int(n));
    produce(0, int(nb)); produce(1, 0); produce(2, int(n_stream)); produce(3, int(n_stream));
    produce(4, int(nv)); produce(5, int(ns)); produce(6, int(n_stream));
    return WORK_CALLED_PRODUCE;
}

bool ofdm_adaptive_rx_hip_impl::stop()
{
    dtl_rx_result r; /* flush: the detector closes an open plateau, incomplete frames are dropped */
    (void)dtl_rx_process_stream(d_rx, nullptr, 0, 0, /*final=*/1, d_stream, &r);
    return true;
}

} // namespace dtl_hip
} // namespace gr
