/* gr::dtl::fec_dec (include/gnuradio/dtl/fec.h:25-31) over the C ABI, for flowgraphs that keep the reference's
 * ofdm_adaptive_fec_decoder block: drop-in for gr::dtl::ldpc_dec (lib/dtl/ldpc_dec.h:19-38).  One codeword per call
 * wastes the device -- the batched path is ofdm_adaptive_rx_hip / include/dtl_blocks.hpp.  Needs gr-dtl's headers. */
#ifndef INCLUDED_DTL_HIP_LDPC_DEC_HIP_H
#define INCLUDED_DTL_HIP_LDPC_DEC_HIP_H

#include <gnuradio/dtl/fec.h>
#include <dtl_rx.h>
#include <hip/hip_runtime_api.h>

#include <stdexcept>
#include <string>
#include <vector>

namespace gr {
namespace dtl {

class ldpc_dec_hip : public fec_dec
{
    dtl_ldpc* d_h = nullptr;
    float* d_llr = nullptr;
    unsigned char* d_bits = nullptr;
    int* d_nit = nullptr;

public:
    ldpc_dec_hip(const std::string& alist_fname, int max_it)
    {
        if (dtl_ldpc_create(alist_fname.c_str(), max_it, 0.0f, &d_h) != DTL_OK) throw std::runtime_error(dtl_last_error());
        if (hipMalloc(reinterpret_cast<void**>(&d_llr), size_t(get_n()) * sizeof(float)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&d_bits), size_t(get_k())) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&d_nit), sizeof(int)) != hipSuccess)
            throw std::runtime_error("ldpc_dec_hip: out of device memory");
    }
    ~ldpc_dec_hip() override
    {
        dtl_ldpc_destroy(d_h);
        (void)hipFree(d_llr); (void)hipFree(d_bits); (void)hipFree(d_nit);
    }
    int decode(const float* in_data, int* nit, unsigned char* out_data) override /* fec.h:28 */
    {
        (void)hipMemcpy(d_llr, in_data, size_t(get_n()) * sizeof(float), hipMemcpyHostToDevice);
        if (dtl_ldpc_decode_batch(d_h, d_llr, 1, d_bits, d_nit, nullptr) != DTL_OK) throw std::runtime_error(dtl_last_error());
        (void)hipMemcpy(out_data, d_bits, size_t(get_k()), hipMemcpyDeviceToHost);
        (void)hipMemcpy(nit, d_nit, sizeof(int), hipMemcpyDeviceToHost);
        return get_k();
    }
    int get_k() override { return dtl_ldpc_get_k(d_h); }
    int get_n() override { return dtl_ldpc_get_n(d_h); }
};

/* make_ldpc_decoders (lib/dtl/ldpc_dec.cc:23-31): index 0 is nullptr ("no FEC") */
inline std::vector<fec_dec::sptr> make_ldpc_decoders_hip(const std::vector<std::string>& alist_fnames)
{
    std::vector<fec_dec::sptr> decoders{ nullptr };
    for (auto& f : alist_fnames) decoders.push_back(std::make_shared<ldpc_dec_hip>(f, 15));
    return decoders;
}

} // namespace dtl
} // namespace gr
#endif
