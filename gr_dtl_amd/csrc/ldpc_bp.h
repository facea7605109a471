// ldpc_bp.h -- device-side LDPC sum-product decoder (kernel K10) for gfx950.
#pragma once

#include "dtl_common.h"
#include "ldpc_code.h"

#include <atomic>

namespace dtl {

// Where one codeword's on-air soft bits live and where its decoded bits go
// (transport-block reassembly of lib/dtl/tb_decoder.cc:143-195 expressed as a gather table).
struct CwDesc {
    uint32_t llr_off0;  // first on-air soft bit (offset into the LLR stream)
    uint32_t llr_off1;  // continuation when the codeword straddles two frames
    uint32_t out_off;   // first decoded bit (1 bit per byte) in the output buffer
    uint16_t len0;      // soft bits taken from llr_off0
    uint16_t len1;      // soft bits taken from llr_off1; sent positions beyond len0+len1 were never received (0)
    uint16_t k_used;    // systematic bits actually sent (k minus shortening)
    uint16_t pad;
};

struct LdpcTables {
    int n, m, k, rank, n_edges, n_chk_pass, n_var_pass;
    const ChkSlot* chk;
    const VarSlot* var;
    const uint16_t* col_edge;
    const uint16_t* row_ptr;
    const uint16_t* edge_var;
    const uint16_t* inv_perm;  // y[v] = -llr[inv_perm[v]]           (ldpc_dec.cc:64-66)
    const uint16_t* sys_src;   // data[i] = estimate[sys_src[i]]     (cldpc::get_systematic_bits)
};

// LDS plan of the node-parallel kernel: byte offsets of the shared index tables and of the
// per-wave slices inside the workgroup's dynamic LDS
struct NpShared {
    int chk_var, var_edge, chk_deg, chk_bits, wave_base, wave_stride;
};

class LdpcDecoder {
public:
    LdpcCode code;
    int max_iterations = 15;  // lib/dtl/ldpc_dec.cc:27
    float sigma = 1.41421356237f; // awgn_bp sigma: never set by gr-dtl; sqrt(2) => rx_lr = exp(llr)
    int arith = 0;            // DTL_ARITH_GR_EXACT (awgn_bp's operations) / DTL_ARITH_FAST (see ldpc_bp.hip)

    ~LdpcDecoder();
    int upload();  // build device tables
    // llr: device float stream; desc==nullptr => codeword i at llr + i*n, all k bits, out at i*k
    // packed: d_bits is a ZEROED, 8-byte aligned buffer that receives 1 bit per decoded bit (LSB first) at bit
    // position out_off instead of one byte per bit
    int decode(const float* d_llr, const CwDesc* d_desc, size_t ncw, uint8_t* d_bits, int32_t* d_nit,
               hipStream_t stream, bool packed = false) const;
    bool packed_output_ok() const { return np_ok_; }
    size_t lds_bytes() const;
    bool uploaded() const { return blob_ != nullptr; }

private:
    LdpcTables t_{};
    void* blob_ = nullptr;
    // node-parallel kernel (normal path)
    NpShared np_sh_{};
    // (variables in ON-AIR numbering: variable i here is the reference's variable permute[i])
    const uint16_t* np_chk_var_ = nullptr;   // m x MAXR: variable per padded message slot (pad: n)
    const uint16_t* np_var_edge_ = nullptr;  // n x 4: message slots of a variable (pad: the slot kept at 1.0)
    const uint8_t* np_chk_deg_ = nullptr;    // m: row weights
    const unsigned long long* np_chk_bits_ = nullptr; // NV x m: rows of H as bitmaps (syndrome check)
    bool np_rows_full_ = false;              // every row has exactly MAXR entries
    int np_maxr_ = 6;
    static constexpr unsigned kNpQueueRing = 256;
    unsigned int* np_queue_ = nullptr;  // ring of ticket counters of the codeword queue: one per launch
    mutable std::atomic<unsigned> np_launch_{ 0 };
    size_t np_lds_ = 0;
    bool np_ok_ = false;
};

} // namespace dtl
