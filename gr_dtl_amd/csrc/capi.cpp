// capi.cpp -- extern "C" boundary of libdtl_hip.so (declared in include/dtl_rx.h).
#include "../../include/dtl_rx.h"

#include "dtl_common.h"
#include "ldpc_bp.h"

#include <fstream>
#include <mutex>
#include <new>
#include <sstream>

namespace dtl {
namespace {
thread_local std::string g_last_error;
}
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* last_error() { return g_last_error.c_str(); }
} // namespace dtl

struct dtl_ldpc {
    dtl::LdpcDecoder dec;
};

static_assert(sizeof(dtl_cw_desc) == sizeof(dtl::CwDesc), "C ABI gather descriptor layout");

extern "C" {

const char* dtl_last_error(void) { return dtl::last_error(); }
const char* dtl_version(void) { return "dtl-mi355x 0.1 (gfx950)"; }

int dtl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dtl_ldpc_create_from_text(const char* text, size_t len, int max_iterations, float sigma,
                              dtl_ldpc** out)
{
    if (!text || !out) { dtl::set_last_error("ldpc_create: null argument"); return DTL_ERR_INVALID_ARG; }
    *out = nullptr;
    auto* h = new (std::nothrow) dtl_ldpc();
    if (!h) return DTL_ERR_CAPACITY;
    std::string err = h->dec.code.parse_alist(std::string(text, len));
    if (!err.empty()) { dtl::set_last_error(err); delete h; return DTL_ERR_INVALID_ARG; }
    h->dec.code.reduce();
    h->dec.code.build_schedules();
    if (max_iterations > 0) h->dec.max_iterations = max_iterations;
    if (sigma > 0) h->dec.sigma = sigma;
    *out = h;
    return DTL_OK;
}

int dtl_ldpc_create(const char* path, int max_iterations, float sigma, dtl_ldpc** out)
{
    if (!path || !out) { dtl::set_last_error("ldpc_create: null argument"); return DTL_ERR_INVALID_ARG; }
    std::ifstream f(path);
    if (!f) { dtl::set_last_error(std::string("cannot open alist ") + path); return DTL_ERR_IO; }
    std::stringstream ss;
    ss << f.rdbuf();
    std::string text = ss.str();
    return dtl_ldpc_create_from_text(text.data(), text.size(), max_iterations, sigma, out);
}

void dtl_ldpc_destroy(dtl_ldpc* h) { delete h; }
int dtl_ldpc_get_n(const dtl_ldpc* h) { return h ? h->dec.code.n : DTL_ERR_INVALID_ARG; }
int dtl_ldpc_get_k(const dtl_ldpc* h) { return h ? h->dec.code.k : DTL_ERR_INVALID_ARG; }

int dtl_ldpc_get_permute(const dtl_ldpc* h, int32_t* out)
{
    if (!h || !out) return DTL_ERR_INVALID_ARG;
    for (int i = 0; i < h->dec.code.n; i++) out[i] = h->dec.code.permute[i];
    return DTL_OK;
}

int dtl_ldpc_encode_host(const dtl_ldpc* h, const uint8_t* data, uint8_t* out)
{
    if (!h || !data || !out) return DTL_ERR_INVALID_ARG;
    h->dec.code.encode(data, out);
    return DTL_OK;
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
    if (!h) { dtl::set_last_error("ldpc_decode: null handle"); return DTL_ERR_INVALID_ARG; }
    if (!h->dec.uploaded()) {
        int rc = ensure_uploaded(h);
        if (rc != DTL_OK) return rc;
    }
    return h->dec.decode(d_llr, reinterpret_cast<const dtl::CwDesc*>(d_desc), ncw, d_bits, d_nit,
                         static_cast<hipStream_t>(stream));
}

int dtl_ldpc_decode_batch(const dtl_ldpc* h, const float* d_llr, size_t ncw, uint8_t* d_bits,
                          int32_t* d_nit, void* stream)
{
    return dtl_ldpc_decode_gather(h, d_llr, nullptr, ncw, d_bits, d_nit, stream);
}

} // extern "C"
