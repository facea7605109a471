/*
 * dtl_rx.h -- C ABI of libdtl_hip.so: the MI355X (gfx950) implementation of gr-dtl's
 * adaptive-OFDM receive path.  Plain pointers and sizes only; every "d_" pointer is DEVICE
 * memory owned by the caller; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All functions return 0 (DTL_OK) or a negative dtl_status; dtl_last_error() gives the text.
 * There is no CPU fallback: without a HIP device every compute entry point fails with
 * DTL_ERR_NO_DEVICE / DTL_ERR_HIP.
 *
 * Each entry point names the reference interface it replaces (paths relative to gr-dtl).
 */
#ifndef DTL_RX_H
#define DTL_RX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum dtl_status {
    DTL_OK = 0,
    DTL_ERR_INVALID_ARG = -1,
    DTL_ERR_IO = -2,
    DTL_ERR_HIP = -3,
    DTL_ERR_TAGS_MISSING = -4,
    DTL_ERR_CAPACITY = -5,
    DTL_ERR_NO_DEVICE = -6,
    DTL_ERR_UNSUPPORTED = -7
};

/* gr::dtl::constellation_type_t, include/gnuradio/dtl/ofdm_adaptive_utils.h:22-28 */
enum dtl_constellation {
    DTL_CNST_UNKNOWN = 0, DTL_CNST_BPSK = 1, DTL_CNST_QPSK = 2, DTL_CNST_PSK8 = 3, DTL_CNST_QAM16 = 4
};

const char* dtl_last_error(void);
const char* dtl_version(void);
/* number of visible HIP devices (0 when none / runtime unusable) */
int dtl_device_count(void);

/* ------------------------------------------------------------------------------------------
 * LDPC decoder  --  replaces gr::dtl::make_ldpc_decoders / ldpc_dec (include/gnuradio/dtl/fec.h:25-37,
 * lib/dtl/ldpc_dec.cc:23-86) and, for the batched form, the per-codeword loop of
 * tb_decoder::decode (lib/dtl/tb_decoder.cc:160-190).
 * ------------------------------------------------------------------------------------------ */
typedef struct dtl_ldpc dtl_ldpc;

/* ldpc_dec::ldpc_dec(alist_fname, max_it): parse the alist, reduce H to systematic form
 * (cldpc), build device tables.  max_iterations: 15 in the reference (ldpc_dec.cc:27).
 * sigma: awgn_bp's noise sigma, which gr-dtl never sets; pass 0 for the documented default
 * sqrt(2) (rx likelihood ratio = exp(llr)). */
int dtl_ldpc_create(const char* alist_path, int max_iterations, float sigma, dtl_ldpc** out);
int dtl_ldpc_create_from_text(const char* alist_text, size_t text_len, int max_iterations,
                              float sigma, dtl_ldpc** out);
void dtl_ldpc_destroy(dtl_ldpc* h);
int dtl_ldpc_get_n(const dtl_ldpc* h);        /* fec_dec::get_n */
int dtl_ldpc_get_k(const dtl_ldpc* h);        /* fec_dec::get_k */
int dtl_ldpc_get_permute(const dtl_ldpc* h, int32_t* out_n); /* cldpc permutation (print_permute hack) */
/* fec_enc::encode (lib/dtl/ldpc_enc.cc:58-68) on HOST buffers: k bits in (1/byte) -> n bits out,
 * on-air order [n-k check bits | k systematic bits].  Test-vector / TX helper. */
int dtl_ldpc_encode_host(const dtl_ldpc* h, const uint8_t* data_k, uint8_t* out_n);

/* fec_dec::decode(const float* in, int* nit, unsigned char* out) for ncw codewords:
 * d_llr  : ncw * n floats, positive = bit 1, on-air order (what ldpc_dec::decode receives)
 * d_bits : ncw * k bytes (0/1), the systematic bits
 * d_nit  : ncw int32 iteration counts (0 = initial hard decision already a codeword) */
int dtl_ldpc_decode_batch(const dtl_ldpc* h, const float* d_llr, size_t ncw, uint8_t* d_bits,
                          int32_t* d_nit, void* stream);

/* Gather-table form used by the transport-block path (tb_decoder::decode): codeword i takes
 * len0 soft bits at llr_off0 then the rest at llr_off1, is padded with the shortening value
 * -15 beyond (n-k)+k_used, and writes k_used decoded bits at out_off. */
typedef struct dtl_cw_desc {
    uint32_t llr_off0, llr_off1, out_off;
    uint16_t len0, k_used;
} dtl_cw_desc;
int dtl_ldpc_decode_gather(const dtl_ldpc* h, const float* d_llr, const dtl_cw_desc* d_desc,
                           size_t ncw, uint8_t* d_bits, int32_t* d_nit, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DTL_RX_H */
