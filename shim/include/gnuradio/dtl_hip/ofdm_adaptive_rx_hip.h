/* GNU Radio block over libdtl_hip.so: the whole receive graph of gr-dtl's ofdm_adaptive_rx hier block
 * (python/dtl/ofdm_adaptive_rx.py:42-59) as ONE block.  Built only where GNU Radio 3.10 is installed; in this
 * repository the source is compiled against tests/cpp/mock_gnuradio/ and EXECUTED on the GPU by
 * tests/test_gr_block_gpu.py (tests/cpp/test_gr_block.cpp drives general_work the way a scheduler thread does). */
#ifndef INCLUDED_DTL_HIP_OFDM_ADAPTIVE_RX_HIP_H
#define INCLUDED_DTL_HIP_OFDM_ADAPTIVE_RX_HIP_H

#include <gnuradio/block.h>
#include <dtl_rx.h>

#include <cstdint>
#include <string>

namespace gr {
namespace dtl_hip {

class ofdm_adaptive_rx_hip : virtual public gr::block
{
public:
    typedef std::shared_ptr<ofdm_adaptive_rx_hip> sptr;

    struct options {
        /* tag keys of the configuration (python/dtl/ofdm_adaptive_config.py:14-16) */
        std::string len_tag_key = "packet_length";
        std::string frame_length_tag_key = "frame_length";
        std::string frame_no_tag_key = "frame_no";
        /* the two switches of ofdm_adaptive_frame_equalizer_vcvc::make; python/dtl/ofdm_receiver.py:172-173 passes (False, 0).
         * propagate_channel_state = true is refused (std::invalid_argument): the channel state at the END of a frame is not
         * exported by the kernels */
        bool propagate_channel_state = false;
        bool propagate_feedback_tags = false;
        /* true: general_work only stages its samples (pinned memory + an asynchronous host-to-device copy on a copy
         * stream) and a worker thread owned by the block decodes whatever has accumulated in ONE dtl_rx_process_stream
         * call -- the scheduler's small buffers become large GPU batches under load, the copy of batch k+1 runs under
         * the decode of batch k, and the outputs of a batch leave with a later general_work call.
         * false: every general_work call decodes its own samples before it returns (deterministic batch boundaries). */
        bool async = true;
        int64_t max_batch_samples = int64_t(1) << 22;   /* capacity of one of the two staging slots */
    };

    /* cfg: the flattened ofdm_adaptive_rx_config (python/dtl/ofdm_adaptive_config.py:10-63), see dtl_rx_cfg; the
     * arrays it points to are copied.  len_tag_key: the packet-length tag put on port 0 (fec_decoder_impl.cc:176). */
    static sptr make(const dtl_rx_cfg& cfg, const std::string& len_tag_key = "packet_length");
    static sptr make(const dtl_rx_cfg& cfg, const options& opt);

    /* returns when everything general_work has accepted so far is decoded and its outputs are queued (async mode; a
     * no-op otherwise).  For tests and orderly shutdown: GNU Radio itself never needs it. */
    virtual void wait_idle() = 0;
};

} // namespace dtl_hip
} // namespace gr
#endif
