// ldpc_code.h -- host-side LDPC code description and device schedule tables.
//
// Replaces what the reference obtains from gr::fec::code::alist + cldpc (GF(2) reduction,
// column permutation, dimension) in lib/dtl/ldpc_dec.cc:34-59 and lib/dtl/ldpc_enc.cc:35-56.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace dtl {

// One lane-slot of the row-aligned check-node schedule (kernel phase "chk").
struct ChkSlot {
    uint16_t edge;       // own edge (check-major index), 0xFFFF = idle lane
    uint16_t row_start;  // first edge of the row
    uint16_t deg;        // row weight
    uint16_t pos;        // position of the own edge inside its row
};

// One lane-slot of the variable-aligned schedule (kernel phase "var").
struct VarSlot {
    uint16_t edge;       // own edge (check-major index), 0xFFFF = idle lane
    uint16_t var;        // variable node
    uint16_t col_start;  // first entry of this variable in col_edge[]
    uint8_t deg;         // column weight
    uint8_t first;       // 1 on the variable's first edge (writes the hard decision)
};

struct LdpcCode {
    int n = 0, m = 0, k = 0, rank = 0, n_edges = 0;
    int max_row_deg = 0, max_col_deg = 0;
    std::vector<int> row_ptr;    // m+1
    std::vector<int> edge_var;   // n_edges, alist row (mlist) order
    std::vector<int> col_ptr;    // n+1
    std::vector<int> col_edge;   // n_edges, alist column (nlist) order -> check-major edge index
    std::vector<int> permute;    // n, column permutation of the systematic form
    std::vector<uint8_t> parity; // rank x k, P of G = [I | P]

    std::vector<ChkSlot> chk_sched;  // n_chk_pass * 64
    std::vector<VarSlot> var_sched;  // n_var_pass * 64
    int n_chk_pass = 0, n_var_pass = 0;

    // Parse alist text ("N M / maxcol maxrow / col wts / row wts / nlist / mlist").
    // Returns empty string on success, else an error message.
    std::string parse_alist(const std::string& text);
    // GF(2) Gauss-Jordan with recorded column swaps (cldpc / GF2Mat::get_G semantics).
    void reduce();
    // Lane schedules for the 64-wide wavefront kernel.
    void build_schedules();
    // cldpc::encode followed by lib/dtl/ldpc_enc.cc:65-67 => [parity | data].
    void encode(const uint8_t* data_k, uint8_t* out_n) const;
};

} // namespace dtl
