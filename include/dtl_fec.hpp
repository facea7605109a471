// dtl_fec.hpp -- C++ drop-in for gr-dtl's FEC objects on top of the C ABI (dtl_rx.h).
//
// Same class names, method signatures and conventions as include/gnuradio/dtl/fec.h:16-37 of the
// reference (namespace gr::dtl; abstract fec_enc / fec_dec; make_ldpc_encoders / make_ldpc_decoders
// returning a vector whose index 0 is nullptr = "no FEC", lib/dtl/ldpc_dec.cc:23-31,
// lib/dtl/ldpc_enc.cc:23-32), so the reference's tb_decoder / fec_utils compile against it unchanged.
// The per-codeword decode() of the reference's interface goes through the device one codeword at a
// time; decode_batch() is the form that uses the GPU properly.
#pragma once

#include "dtl_rx.h"

#include <hip/hip_runtime_api.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace gr {
namespace dtl {

class fec_enc
{
public:
    typedef std::shared_ptr<fec_enc> sptr;
    virtual ~fec_enc() = default;
    virtual void encode(const unsigned char* in_data, int len, unsigned char* out_data) = 0;
    virtual int get_k() = 0;
    virtual int get_n() = 0;
};

class fec_dec
{
public:
    typedef std::shared_ptr<fec_dec> sptr;
    virtual ~fec_dec() = default;
    virtual int decode(const float* in_data, int* nit, unsigned char* out_data) = 0;
    virtual int get_k() = 0;
    virtual int get_n() = 0;
};

namespace detail {
inline void check(int rc)
{
    if (rc != DTL_OK) throw std::runtime_error(std::string("libdtl_hip: ") + dtl_last_error());
}
inline void hip_check(hipError_t e)
{
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e));
}
} // namespace detail

// gr::dtl::ldpc_enc (lib/dtl/ldpc_enc.h:19-36)
class ldpc_enc : public fec_enc
{
    dtl_ldpc* d_h = nullptr;
    std::vector<unsigned char> d_in;

public:
    explicit ldpc_enc(const std::string& alist_fname)
    {
        detail::check(dtl_ldpc_create(alist_fname.c_str(), 15, 0.0f, &d_h));
        d_in.resize(get_k());
    }
    ~ldpc_enc() override { dtl_ldpc_destroy(d_h); }
    void encode(const unsigned char* in_data, int len, unsigned char* out_data) override
    {
        std::fill(d_in.begin(), d_in.end(), 0);
        std::copy(in_data, in_data + std::min<int>(len, get_k()), d_in.begin());
        detail::check(dtl_ldpc_encode_host(d_h, d_in.data(), out_data));
    }
    int get_k() override { return dtl_ldpc_get_k(d_h); }
    int get_n() override { return dtl_ldpc_get_n(d_h); }
};

// gr::dtl::ldpc_dec (lib/dtl/ldpc_dec.h:19-38)
class ldpc_dec : public fec_dec
{
    dtl_ldpc* d_h = nullptr;
    float* d_llr = nullptr;
    unsigned char* d_bits = nullptr;
    int* d_nit = nullptr;
    size_t d_cap = 0;

    void reserve(size_t ncw)
    {
        if (ncw <= d_cap) return;
        release();
        detail::hip_check(hipMalloc(reinterpret_cast<void**>(&d_llr), ncw * get_n() * sizeof(float)));
        detail::hip_check(hipMalloc(reinterpret_cast<void**>(&d_bits), ncw * get_k()));
        detail::hip_check(hipMalloc(reinterpret_cast<void**>(&d_nit), ncw * sizeof(int)));
        d_cap = ncw;
    }
    void release()
    {
        if (d_llr) (void)hipFree(d_llr);
        if (d_bits) (void)hipFree(d_bits);
        if (d_nit) (void)hipFree(d_nit);
        d_llr = nullptr; d_bits = nullptr; d_nit = nullptr; d_cap = 0;
    }

public:
    ldpc_dec(const std::string& alist_fname, int max_it)
    {
        detail::check(dtl_ldpc_create(alist_fname.c_str(), max_it, 0.0f, &d_h));
    }
    ~ldpc_dec() override
    {
        release();
        dtl_ldpc_destroy(d_h);
    }
    // fec_dec::decode: in_data = n soft bits (positive = 1, on-air order), out_data = k bits, *nit = iterations
    int decode(const float* in_data, int* nit, unsigned char* out_data) override
    {
        decode_batch(in_data, 1, nit, out_data);
        return get_k();
    }
    // ncw codewords back to back (host buffers)
    void decode_batch(const float* in_data, size_t ncw, int* nit, unsigned char* out_data)
    {
        reserve(ncw);
        detail::hip_check(hipMemcpy(d_llr, in_data, ncw * get_n() * sizeof(float), hipMemcpyHostToDevice));
        detail::check(dtl_ldpc_decode_batch(d_h, d_llr, ncw, d_bits, d_nit, nullptr));
        detail::hip_check(hipMemcpy(out_data, d_bits, ncw * get_k(), hipMemcpyDeviceToHost));
        detail::hip_check(hipMemcpy(nit, d_nit, ncw * sizeof(int), hipMemcpyDeviceToHost));
    }
    int get_k() override { return dtl_ldpc_get_k(d_h); }
    int get_n() override { return dtl_ldpc_get_n(d_h); }
    // the C-ABI handle, for the batched device entry points (dtl_ldpc_decode_batch / _gather)
    const dtl_ldpc* handle() const { return d_h; }
    // DTL_ARITH_GR_EXACT (default: the reference decoder's operations) / DTL_ARITH_FAST
    void set_arith(int arith) { detail::check(dtl_ldpc_set_arith(d_h, arith)); }
};

inline std::vector<fec_enc::sptr> make_ldpc_encoders(const std::vector<std::string>& alist_fnames)
{
    std::vector<fec_enc::sptr> encoders{ nullptr };
    for (auto& fname : alist_fnames) encoders.push_back(std::make_shared<ldpc_enc>(fname));
    return encoders;
}

inline std::vector<fec_dec::sptr> make_ldpc_decoders(const std::vector<std::string>& alist_fnames)
{
    std::vector<fec_dec::sptr> decoders{ nullptr };
    for (auto& fname : alist_fnames) decoders.push_back(std::make_shared<ldpc_dec>(fname, 15)); // ldpc_dec.cc:27
    return decoders;
}

} // namespace dtl
} // namespace gr
