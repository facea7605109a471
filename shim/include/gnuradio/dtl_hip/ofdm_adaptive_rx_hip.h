/* GNU Radio block over libdtl_hip.so: the whole receive graph of gr-dtl's ofdm_adaptive_rx hier block
 * (python/dtl/ofdm_adaptive_rx.py:42-59) as ONE block.  Not compiled in this repository (needs GNU Radio 3.10). */
#ifndef INCLUDED_DTL_HIP_OFDM_ADAPTIVE_RX_HIP_H
#define INCLUDED_DTL_HIP_OFDM_ADAPTIVE_RX_HIP_H

#include <gnuradio/block.h>
#include <dtl_rx.h>

#include <string>

namespace gr {
namespace dtl_hip {

class ofdm_adaptive_rx_hip : virtual public gr::block
{
public:
    typedef std::shared_ptr<ofdm_adaptive_rx_hip> sptr;
    /* cfg: the flattened ofdm_adaptive_rx_config (python/dtl/ofdm_adaptive_config.py:10-63), see dtl_rx_cfg; the
     * arrays it points to are copied.  len_tag_key: the packet-length tag put on port 0 (fec_decoder_impl.cc:176). */
    static sptr make(const dtl_rx_cfg& cfg, const std::string& len_tag_key = "len_tag_key");
};

} // namespace dtl_hip
} // namespace gr
#endif
