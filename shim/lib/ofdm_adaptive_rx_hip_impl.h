#ifndef INCLUDED_DTL_HIP_OFDM_ADAPTIVE_RX_HIP_IMPL_H
#define INCLUDED_DTL_HIP_OFDM_ADAPTIVE_RX_HIP_IMPL_H

#include <gnuradio/dtl_hip/ofdm_adaptive_rx_hip.h>
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace gr {
namespace dtl_hip {

class ofdm_adaptive_rx_hip_impl : public ofdm_adaptive_rx_hip
{
    /* what one decoded batch hands back to general_work */
    struct frame_out {
        dtl_rx_frame_info fi;
    };
    struct batch_out {
        int64_t end = 0;                    /* absolute sample index behind the batch */
        std::vector<unsigned char> bytes;   /* port 0 */
        std::vector<dtl_rx_tb_info> tbs;
        std::vector<frame_out> frames;      /* every trigger the demux visited */
        std::vector<gr_complex> hard, soft; /* ports 4, 5: frame_length x fft_len per payload frame */
        std::vector<int64_t> raw, fixed;
        std::vector<float> angle;
        bool final = false;
    };
    struct slot {
        gr_complex* pinned = nullptr;       /* page-locked staging: a pageable hipMemcpyAsync pins per call */
        gr_complex* dev = nullptr;
        int64_t fill = 0;                   /* samples staged (copies issued on d_copy) */
    };

    options d_opt;
    dtl_rx* d_rx = nullptr;
    hipStream_t d_stream = nullptr, d_copy = nullptr;
    hipEvent_t d_copied = nullptr;
    int d_fft_len, d_frame_length, d_frame_samples, d_has_fec;
    pmt::pmt_t d_len_key, d_frame_len_key, d_frame_no_key;
    slot d_slot[2];
    int d_fill = 0;                         /* the slot general_work stages into */
    void* d_out_pinned = nullptr;           /* device-to-host staging of a batch's outputs */
    size_t d_out_pinned_cap = 0;
    int64_t d_pos = 0;                      /* samples handed to the receiver so far */
    /* worker (async) */
    std::thread d_worker;
    std::mutex d_mu;
    std::condition_variable d_cv_work, d_cv_space, d_cv_idle;
    bool d_quit = false, d_busy = false, d_flush = false;
    std::deque<batch_out> d_results;
    std::string d_error;                    /* a failed batch: thrown from the next general_work (reference: exceptions out of work()) */
    /* per-sample ports (2: fixed triggers, 3: raw triggers, 6: held fine frequency) trail the decoded input by one
     * frame: a trigger is known when its plateau has ended and frame_detect_bb has seen the samples behind it */
    std::deque<int64_t> d_raw, d_fixed;
    std::deque<float> d_angle;
    int64_t d_emitted = 0, d_decoded_end = 0;
    float d_held = 0.0f;
    bool d_final_seen = false;
    /* output that did not fit the scheduler's buffers waits here; tags wait with the item they sit on */
    std::deque<unsigned char> d_pending_bytes;
    std::deque<gr_complex> d_pending_hard, d_pending_soft;
    std::deque<tag_t> d_pending_tags[7];
    uint64_t d_queued[7] = { 0, 0, 0, 0, 0, 0, 0 };   /* items ever queued per port (absolute offsets of pending tags) */
    long d_lost_total = 0, d_frames_count = 0;

    void ensure_out_pinned(size_t bytes);
    void stage(const gr_complex* in, int64_t n);
    bool take_batch(int& b, int64_t& n);            /* under d_mu */
    void run_batch(int b, int64_t n, bool final);
    void worker_loop();
    void drain_results();
    void queue_batch(const batch_out& r);
    void queue_tag(int port, uint64_t offset, const pmt::pmt_t& key, const pmt::pmt_t& value);
    int emit_tags(int port, uint64_t upto);

public:
    ofdm_adaptive_rx_hip_impl(const dtl_rx_cfg& cfg, const options& opt);
    ~ofdm_adaptive_rx_hip_impl() override;
    void forecast(int noutput_items, gr_vector_int& ninput_items_required) override;
    bool stop() override;
    void wait_idle() override;
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items) override;
};

} // namespace dtl_hip
} // namespace gr
#endif
