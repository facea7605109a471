#ifndef INCLUDED_DTL_HIP_OFDM_ADAPTIVE_RX_HIP_IMPL_H
#define INCLUDED_DTL_HIP_OFDM_ADAPTIVE_RX_HIP_IMPL_H

#include <gnuradio/dtl_hip/ofdm_adaptive_rx_hip.h>
#include <hip/hip_runtime_api.h>

#include <deque>
#include <vector>

namespace gr {
namespace dtl_hip {

class ofdm_adaptive_rx_hip_impl : public ofdm_adaptive_rx_hip
{
    dtl_rx* d_rx = nullptr;
    hipStream_t d_stream = nullptr;
    void* d_iq = nullptr;           /* device staging of one work() buffer */
    size_t d_iq_cap = 0;
    void* d_pinned = nullptr;       /* page-locked host staging (a pageable hipMemcpyAsync pins per call) */
    int d_fft_len, d_frame_length, d_frame_samples;
    pmt::pmt_t d_len_key;
    /* per-sample ports (2: fixed triggers, 3: raw triggers, 6: held fine frequency) trail the input by one frame:
     * a trigger is known when its plateau has ended and frame_detect_bb has seen the samples behind it */
    std::deque<int64_t> d_raw, d_fixed;
    std::deque<float> d_angle;
    int64_t d_emitted = 0;
    float d_held = 0.0f;
    /* output of a call that did not fit the scheduler's buffers waits here */
    std::vector<unsigned char> d_pending_bytes;
    std::vector<gr_complex> d_pending_hard, d_pending_soft;

    void ensure(size_t n_samples);
    void publish_messages(const dtl_rx_result& r);

public:
    ofdm_adaptive_rx_hip_impl(const dtl_rx_cfg& cfg, const std::string& len_tag_key);
    ~ofdm_adaptive_rx_hip_impl() override;
    void forecast(int noutput_items, gr_vector_int& ninput_items_required) override;
    bool stop() override;
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items) override;
};

} // namespace dtl_hip
} // namespace gr
#endif
