// ldpc_code.cpp -- alist parsing, systematic-form reduction and wavefront schedules.
#include "ldpc_code.h"

#include <algorithm>
#include <sstream>

namespace dtl {

std::string LdpcCode::parse_alist(const std::string& text)
{
    std::istringstream is(text);
    int big_n = 0, big_m = 0;
    if (!(is >> n >> m >> big_n >> big_m) || n <= 0 || m <= 0) return "alist: bad header";
    std::vector<int> col_w(n), row_w(m);
    for (auto& w : col_w)
        if (!(is >> w) || w < 0) return "alist: bad column weights";
    for (auto& w : row_w)
        if (!(is >> w) || w < 0) return "alist: bad row weights";
    std::string line;
    std::getline(is, line); // remainder of the weights line
    auto read_list = [&](int count, const std::vector<int>& w, std::vector<std::vector<int>>& dst) {
        dst.assign(count, {});
        for (int i = 0; i < count; i++) {
            if (!std::getline(is, line)) return false;
            std::istringstream ls(line);
            dst[i].resize(w[i]);
            for (int e = 0; e < w[i]; e++)
                if (!(ls >> dst[i][e])) return false; // zero padding after w[i] entries is ignored
        }
        return true;
    };
    std::vector<std::vector<int>> nlist, mlist;
    if (!read_list(n, col_w, nlist)) return "alist: bad nlist";
    if (!read_list(m, row_w, mlist)) return "alist: bad mlist";

    row_ptr.assign(m + 1, 0);
    for (int r = 0; r < m; r++) row_ptr[r + 1] = row_ptr[r] + row_w[r];
    n_edges = row_ptr[m];
    if (n_edges >= 0xFFFF || n >= 0xFFFF) return "alist: code too large for 16-bit schedules";
    edge_var.resize(n_edges);
    max_row_deg = max_col_deg = 0;
    for (int r = 0; r < m; r++) {
        max_row_deg = std::max(max_row_deg, row_w[r]);
        for (int e = 0; e < row_w[r]; e++) {
            int v = mlist[r][e] - 1;
            if (v < 0 || v >= n) return "alist: mlist entry out of range";
            edge_var[row_ptr[r] + e] = v;
        }
    }
    col_ptr.assign(n + 1, 0);
    for (int v = 0; v < n; v++) col_ptr[v + 1] = col_ptr[v] + col_w[v];
    if (col_ptr[n] != n_edges) return "alist: nlist/mlist edge counts differ";
    col_edge.assign(n_edges, -1);
    for (int v = 0; v < n; v++) {
        max_col_deg = std::max(max_col_deg, col_w[v]);
        for (int i = 0; i < col_w[v]; i++) {
            int c = nlist[v][i] - 1;
            if (c < 0 || c >= m) return "alist: nlist entry out of range";
            auto b = edge_var.begin() + row_ptr[c], e = edge_var.begin() + row_ptr[c + 1];
            auto it = std::find(b, e, v);
            if (it == e) return "alist: nlist and mlist disagree";
            col_edge[col_ptr[v] + i] = int(it - edge_var.begin());
        }
    }
    if (max_row_deg > 64) return "alist: row weight > 64 unsupported";
    return {};
}

namespace {
// dense GF(2) matrix, rows packed in 64-bit words
struct BitMat {
    int rows, cols, words;
    std::vector<uint64_t> d;
    BitMat(int r, int c) : rows(r), cols(c), words((c + 63) / 64), d(size_t(r) * words, 0) {}
    bool get(int r, int c) const { return (d[size_t(r) * words + (c >> 6)] >> (c & 63)) & 1u; }
    void flip(int r, int c) { d[size_t(r) * words + (c >> 6)] ^= uint64_t(1) << (c & 63); }
    void set(int r, int c) { d[size_t(r) * words + (c >> 6)] |= uint64_t(1) << (c & 63); }
    void xor_row(int dst, int src)
    {
        for (int w = 0; w < words; w++) d[size_t(dst) * words + w] ^= d[size_t(src) * words + w];
    }
    void swap_cols(int a, int b)
    {
        for (int r = 0; r < rows; r++)
            if (get(r, a) != get(r, b)) { flip(r, a); flip(r, b); }
    }
};
} // namespace

void LdpcCode::reduce()
{
    BitMat g(m, n);
    for (int r = 0; r < m; r++)
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; e++) g.set(r, edge_var[e]);
    permute.resize(n);
    for (int c = 0; c < n; c++) permute[c] = c;
    rank = 0;
    // Pivot on (diag, diag): first non-zero column at or right of diag in row diag is swapped
    // in; a row without pivot is rotated to the bottom and the limit shrinks.
    int diag = 0, limit = m;
    while (diag < limit) {
        int pivot = -1;
        for (int c = diag; c < n; c++)
            if (g.get(diag, c)) { pivot = c; break; }
        if (pivot >= 0) {
            rank++;
            if (pivot != diag) {
                g.swap_cols(diag, pivot);
                std::swap(permute[diag], permute[pivot]);
            }
            for (int r = 0; r < m; r++)
                if (r != diag && g.get(r, diag)) g.xor_row(r, diag);
            diag++;
        } else {
            std::vector<uint64_t> keep(g.d.begin() + size_t(diag) * g.words,
                                       g.d.begin() + size_t(diag + 1) * g.words);
            for (int r = diag; r < limit - 1; r++)
                std::copy(g.d.begin() + size_t(r + 1) * g.words, g.d.begin() + size_t(r + 2) * g.words,
                          g.d.begin() + size_t(r) * g.words);
            std::copy(keep.begin(), keep.end(), g.d.begin() + size_t(limit - 1) * g.words);
            limit--;
        }
    }
    k = n - rank;
    parity.assign(size_t(rank) * k, 0);
    for (int i = 0; i < rank; i++)
        for (int j = 0; j < k; j++) parity[size_t(i) * k + j] = g.get(i, rank + j);
}

void LdpcCode::build_schedules()
{
    // Check phase: whole rows are packed greedily into 64-lane passes so that every lane of
    // a row sits in the same wave instruction (in-place update of the row is then safe).
    chk_sched.clear();
    auto pad_pass = [](auto& sched, auto idle) {
        while (sched.size() % 64) sched.push_back(idle);
    };
    const ChkSlot chk_idle{ 0xFFFF, 0, 0, 0 };
    for (int r = 0; r < m; r++) {
        int deg = row_ptr[r + 1] - row_ptr[r];
        if (int(chk_sched.size() % 64) + deg > 64) pad_pass(chk_sched, chk_idle);
        for (int j = 0; j < deg; j++)
            chk_sched.push_back({ uint16_t(row_ptr[r] + j), uint16_t(row_ptr[r]), uint16_t(deg), uint16_t(j) });
    }
    pad_pass(chk_sched, chk_idle);
    n_chk_pass = int(chk_sched.size() / 64);

    // Variable phase: whole variables per pass, same reason.
    var_sched.clear();
    const VarSlot var_idle{ 0xFFFF, 0, 0, 0, 0 };
    for (int v = 0; v < n; v++) {
        int deg = col_ptr[v + 1] - col_ptr[v];
        if (int(var_sched.size() % 64) + deg > 64) pad_pass(var_sched, var_idle);
        for (int i = 0; i < deg; i++)
            var_sched.push_back({ uint16_t(col_edge[col_ptr[v] + i]), uint16_t(v), uint16_t(col_ptr[v]),
                                  uint8_t(deg), uint8_t(i == 0) });
    }
    pad_pass(var_sched, var_idle);
    n_var_pass = int(var_sched.size() / 64);
}

void LdpcCode::encode(const uint8_t* data, uint8_t* out) const
{
    for (int i = 0; i < rank; i++) {
        unsigned acc = 0;
        const uint8_t* row = &parity[size_t(i) * k];
        for (int j = 0; j < k; j++) acc ^= unsigned(row[j] & data[j] & 1u);
        out[i] = uint8_t(acc);
    }
    for (int j = 0; j < k; j++) out[rank + j] = data[j] & 1u;
}

} // namespace dtl
