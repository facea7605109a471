// tx_pipeline.cpp -- host side of the batched transmitter: configuration tables, the framing state
// machine of ofdm_adaptive_fec_frame_bvb / ofdm_adaptive_frame_bb evaluated on lengths, header bits.
#include "tx_pipeline.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace dtl {

#define DTL_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != DTL_OK) return _rc; \
    } while (0)

template <typename T>
static int upload_vec(HostBuf& hb, DevBuf& db, const std::vector<T>& v, hipStream_t st)
{
    const size_t bytes = v.size() * sizeof(T);
    if (!bytes) return DTL_OK;
    DTL_TRY(hb.ensure(bytes));
    DTL_TRY(db.ensure(bytes));
    std::memcpy(hb.p, v.data(), bytes);
    DTL_HIP_CHECK(hipMemcpyAsync(db.p, hb.p, bytes, hipMemcpyHostToDevice, st));
    return DTL_OK;
}

static int bits_per_symbol(int cnst) { return (cnst >= 1 && cnst <= 4) ? cnst : 0; }

// pack_crc (packet_header.cc:93-105, including its `i*len+j < header_len` bound) followed by
// crc(16, 0x1021, 0xFFFF, 0, refin = false, refout = true)
static unsigned header_crc16(const uint8_t* bits, int header_len)
{
    const int ln = (header_len - 16) / 8 + (((header_len - 16) % 8) ? 1 : 0);
    unsigned reg = 0xFFFF;
    for (int i = 0; i < ln; i++) {
        unsigned byte = 0;
        for (int j = 0; j < 8; j++)
            if (i * ln + j < header_len) byte = ((byte << 1) | (bits[i * 8 + j] & 1u)) & 0xFFu;
        for (int b = 7; b >= 0; b--) {
            const unsigned msb = (reg >> 15) & 1u;
            reg = (reg << 1) & 0xFFFFu;
            if (msb ^ ((byte >> b) & 1u)) reg ^= 0x1021u;
        }
    }
    unsigned rev = 0;
    for (int i = 0; i < 16; i++) if (reg & (1u << i)) rev |= 1u << (15 - i);
    return rev;
}

void format_header_bits(const TxFrameInfo& f, bool has_fec, int header_len, uint8_t* out)
{
    std::memset(out, 0, size_t(header_len));
    auto put = [&](int pos, unsigned val, int nbits) {
        for (int i = 0; i < nbits; i++)
            if (pos + i < header_len) out[pos + i] = uint8_t((val >> i) & 1u);
        return pos + nbits;
    };
    int k = 0;
    k = put(k, unsigned(f.payload_len) & 0xFFFu, 12);
    k = put(k, unsigned(f.frame_no) & 0xFFFu, 12);
    k = put(k, unsigned(f.cnst) & 0xFu, 4);
    k = put(k, unsigned(f.fb_cnst) & 0xFu, 4);
    if (has_fec) {
        put(k + 0, unsigned(f.tb_no), 12);
        put(k + 12, unsigned(f.fec_fb), 4);
        put(k + 16, unsigned(f.tb_offset), 12);
        put(k + 28, unsigned(f.fec), 4);
        put(k + 32, unsigned(f.tb_payload), 16);
        k += 48;
    }
    put(k, header_crc16(out, header_len), 16);
}

int Transmitter::init()
{
    const int N = cfg.fft_len;
    if (N < 64 || N > 1024 || (N & (N - 1))) { set_last_error("fft_len must be a power of two in [64,1024]"); return DTL_ERR_INVALID_ARG; }
    if (cfg.cp_len < 0 || cfg.cp_len > N) { set_last_error("bad cp_len"); return DTL_ERR_INVALID_ARG; }
    if (cfg.frame_length < 1) { set_last_error("bad frame_length"); return DTL_ERR_INVALID_ARG; }
    if (cfg.occupied_carriers.empty() || cfg.occupied_carriers[0].empty()) { set_last_error("occupied_carriers empty"); return DTL_ERR_INVALID_ARG; }
    if (int(cfg.sync_word1.size()) != N || int(cfg.sync_word2.size()) != N) {
        set_last_error("Length of sync sequence(s) must be FFT length."); // ofdm_transmitter.py:57-59
        return DTL_ERR_INVALID_ARG;
    }
    if (cfg.scramble_header) { set_last_error("tx: header scrambling is not implemented"); return DTL_ERR_UNSUPPORTED; }
    for (auto& set : cfg.occupied_carriers)
        for (int c : set) {
            const int idx = c < 0 ? c + N : c;
            if (idx < 0 || idx >= N) { set_last_error("data carrier index out of bounds"); return DTL_ERR_INVALID_ARG; }
        }
    const size_t np = cfg.pilot_carriers.empty() ? 0 : cfg.pilot_carriers[0].size();
    for (auto& set : cfg.pilot_carriers)
        if (set.size() != np) { set_last_error("pilot carrier sets must have equal size"); return DTL_ERR_INVALID_ARG; }
    for (auto& set : cfg.pilot_symbols)
        if (set.size() != np) { set_last_error("pilot symbol sets must match the pilot carriers"); return DTL_ERR_INVALID_ARG; }
    if (!cfg.pilot_carriers.empty() && cfg.pilot_symbols.empty()) { set_last_error("pilot_symbols empty"); return DTL_ERR_INVALID_ARG; }
    { int rc = cfg.check_header_plan(); if (rc != DTL_OK) return rc; }   // the allocator would start the payload inside the header symbols
    capacity_ = 0;
    for (int i = 0; i < cfg.frame_length; i++) capacity_ += int(cfg.occupied_carriers[i % cfg.occupied_carriers.size()].size());
    codes_.clear();
    codes_.emplace_back(); // index 0: "no FEC" (ldpc_enc.cc:25)
    for (auto& text : cfg.alist_texts) {
        LdpcCode c;
        std::string err = c.parse_alist(text);
        if (!err.empty()) { set_last_error(err); return DTL_ERR_INVALID_ARG; }
        c.reduce();
        codes_.push_back(std::move(c));
    }
    if (cfg.has_fec && codes_.size() < 2) { set_last_error("Misconfiguration: encoder not found!"); return DTL_ERR_INVALID_ARG; }
    return DTL_OK;
}

int Transmitter::upload_tables()
{
    const int N = cfg.fft_len;
    auto shifted = [&](int c) { int idx = c < 0 ? c + N : c; return (idx + N / 2) % N; };
    std::vector<uint8_t> blob;
    auto put = [&](const void* p, size_t bytes) {
        size_t off = (blob.size() + 15) & ~size_t(15);
        blob.resize(off + bytes);
        if (bytes) std::memcpy(blob.data() + off, p, bytes);
        return off;
    };
    std::vector<cf32> tw(N / 2);
    for (int k = 0; k < N / 2; k++) {
        const double ang = 2.0 * M_PI * k / N;
        tw[k] = { float(std::cos(ang)), float(std::sin(ang)) };
    }
    std::vector<uint16_t> br(N);
    int logn = 0;
    while ((1 << logn) < N) logn++;
    for (int i = 0; i < N; i++) {
        int r = 0;
        for (int b = 0; b < logn; b++) if (i & (1 << b)) r |= 1 << (logn - 1 - b);
        br[i] = uint16_t(r);
    }
    const int n_occ_sets = int(cfg.occupied_carriers.size());
    std::vector<int16_t> occ_rank(size_t(n_occ_sets) * N, -1), occ_len(n_occ_sets);
    for (int s = 0; s < n_occ_sets; s++) {
        occ_len[s] = int16_t(cfg.occupied_carriers[s].size());
        // a carrier listed twice keeps the LAST data symbol written to it (allocator order)
        for (size_t k = 0; k < cfg.occupied_carriers[s].size(); k++) occ_rank[size_t(s) * N + shifted(cfg.occupied_carriers[s][k])] = int16_t(k);
    }
    const int n_pilot_sets = int(cfg.pilot_carriers.size());
    const int n_pilots = n_pilot_sets ? int(cfg.pilot_carriers[0].size()) : 0;
    const int n_psym = int(cfg.pilot_symbols.size());
    std::vector<int16_t> pil_rank(size_t(std::max(n_pilot_sets, 1)) * N, -1);
    for (int s = 0; s < n_pilot_sets; s++)
        for (int k = 0; k < n_pilots; k++)
            if (cfg.pilot_carriers[s][k] != DTL_NO_CARRIER) pil_rank[size_t(s) * N + shifted(cfg.pilot_carriers[s][k])] = int16_t(k);
    std::vector<cf32> psyms(size_t(std::max(n_psym, 1)) * std::max(n_pilots, 1), cf32{ 0, 0 });
    for (int s = 0; s < n_psym; s++)
        for (int k = 0; k < n_pilots; k++) psyms[size_t(s) * n_pilots + k] = cfg.pilot_symbols[s][k];
    // constellation tables (gr::digital constellation_{bpsk,qpsk,8psk,16qam}; dtl QPSK x0.5)
    std::vector<cf32> pts(5 * 16, cf32{ 0, 0 });
    pts[1 * 16 + 0] = { -1, 0 }; pts[1 * 16 + 1] = { 1, 0 };
    {
        const float s = float(double(float(0.707107)) * 0.5); // SQRT_TWO macro, lib/dtl/constellation.cc:19-23
        pts[2 * 16 + 0] = { -s, -s }; pts[2 * 16 + 1] = { s, -s }; pts[2 * 16 + 2] = { -s, s }; pts[2 * 16 + 3] = { s, s };
    }
    {
        const float angle = float(M_PI / 8.0);
        const int mult[8] = { 1, 7, 15, 9, 3, 5, 13, 11 };
        for (int i = 0; i < 8; i++) pts[3 * 16 + i] = { float(std::cos(mult[i] * angle)), float(std::sin(mult[i] * angle)) };
    }
    {
        const float level = std::sqrt(float(0.1));
        for (int i = 0; i < 16; i++)
            pts[4 * 16 + i] = { ((i & 2) ? 3 : 1) * level * ((i & 4) ? -1.f : 1.f), ((i & 1) ? 3 : 1) * level * ((i & 8) ? 1.f : -1.f) };
    }
    const size_t o_tw = put(tw.data(), tw.size() * sizeof(cf32));
    const size_t o_br = put(br.data(), br.size() * 2);
    const size_t o_or = put(occ_rank.data(), occ_rank.size() * 2);
    const size_t o_ol = put(occ_len.data(), occ_len.size() * 2);
    const size_t o_pr = put(pil_rank.data(), pil_rank.size() * 2);
    const size_t o_ps = put(psyms.data(), psyms.size() * sizeof(cf32));
    const size_t o_s1 = put(cfg.sync_word1.data(), size_t(N) * sizeof(cf32));
    const size_t o_s2 = put(cfg.sync_word2.data(), size_t(N) * sizeof(cf32));
    const size_t o_pt = put(pts.data(), pts.size() * sizeof(cf32));
    // generator parts, bit-packed: row i of P as ceil(k/64) words, bit j of word w = P[i][64 w + j]
    std::vector<size_t> o_par(codes_.size(), 0);
    for (size_t c = 1; c < codes_.size(); c++) {
        const LdpcCode& code = codes_[c];
        const int words = (code.k + 63) / 64;
        std::vector<unsigned long long> packed(size_t(code.rank) * words, 0ull);
        for (int i = 0; i < code.rank; i++)
            for (int j = 0; j < code.k; j++)
                if (code.parity[size_t(i) * code.k + j] & 1u) packed[size_t(i) * words + j / 64] |= 1ull << (j % 64);
        o_par[c] = put(packed.data(), packed.size() * 8);
    }
    DTL_TRY(tab_blob_.ensure(blob.size()));
    DTL_HIP_CHECK(hipMemcpy(tab_blob_.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    auto* b = tab_blob_.as<uint8_t>();
    tab_.fft_len = N; tab_.cp_len = cfg.cp_len; tab_.frame_length = cfg.frame_length; tab_.header_len = cfg.header_len();
    tab_.n_occ_sets = n_occ_sets; tab_.n_pilot_sets = n_pilot_sets; tab_.n_pilots = n_pilots; tab_.n_pilot_sym_sets = n_psym;
    tab_.frame_capacity = capacity_; tab_.hdr_bits = hdr_bits();
    tab_.twiddle = reinterpret_cast<const cf32*>(b + o_tw);
    tab_.bitrev = reinterpret_cast<const uint16_t*>(b + o_br);
    tab_.occ_rank = reinterpret_cast<const int16_t*>(b + o_or);
    tab_.occ_len = reinterpret_cast<const int16_t*>(b + o_ol);
    tab_.pil_rank = reinterpret_cast<const int16_t*>(b + o_pr);
    tab_.pilot_syms = reinterpret_cast<const cf32*>(b + o_ps);
    tab_.sync1 = reinterpret_cast<const cf32*>(b + o_s1);
    tab_.sync2 = reinterpret_cast<const cf32*>(b + o_s2);
    tab_.cnst_pts = reinterpret_cast<const cf32*>(b + o_pt);
    code_tab_.assign(codes_.size(), TxCodeTables{});
    for (size_t c = 1; c < codes_.size(); c++) {
        code_tab_[c].n = codes_[c].n; code_tab_[c].k = codes_[c].k; code_tab_[c].rank = codes_[c].rank;
        code_tab_[c].words = (codes_[c].k + 63) / 64;
        code_tab_[c].parity = reinterpret_cast<const unsigned long long*>(b + o_par[c]);
    }
    uploaded_ = true;
    return DTL_OK;
}

// The framing state machine on lengths.  FEC: ofdm_adaptive_fec_frame_bvb_impl.cc:284-565 for one PDU and a
// constant MCS (each transport block = user bytes + CRC-32, tb_encoder split into ncws codewords, streamed
// into frames byte-aligned, a new block starting on a symbol boundary inside a frame).  No FEC:
// ofdm_adaptive_frame_bb_impl.cc:131-163 (one CRC-protected byte string per frame).
int Transmitter::plan(int64_t n_user, int cnst, int fec, int first_frame_no, int first_tb_no)
{
    frames.clear(); tbs_.clear(); segs_.clear(); tframes_.clear();
    coded_total_ = 0;
    const int bps = bits_per_symbol(cnst), cap = capacity_;
    if (!bps) { set_last_error("Unknown constellation"); return DTL_ERR_INVALID_ARG; }
    if (!cfg.has_fec) {
        const int per = cap * bps / 8 - 4;
        if (per <= 0) { set_last_error("tx: frame too small for a CRC-32"); return DTL_ERR_INVALID_ARG; }
        int f = 0;
        for (int64_t lo = 0; lo < n_user; lo += per, f++) {
            const uint32_t len = uint32_t(std::min<int64_t>(per, n_user - lo));
            TxTb tb{ uint32_t(lo), len, uint32_t(coded_total_), uint32_t(per + 4) * 8u, 0, 0 };
            tbs_.push_back(tb);
            segs_.push_back({ tb.coded_off, std::min<uint32_t>(tb.coded_len, uint32_t(cap) * bps), 0, 0 });
            tframes_.push_back({ int(segs_.size()) - 1, 1, cnst, 0 });
            coded_total_ += tb.coded_len;
            TxFrameInfo fi{};
            fi.payload_len = int(len) + 4; fi.frame_no = (first_frame_no + f) & 0xFFF; fi.cnst = cnst;
            frames.push_back(fi);
        }
        return DTL_OK;
    }
    if (fec <= 0 || fec >= int(codes_.size())) { set_last_error("tx: unknown FEC scheme"); return DTL_ERR_INVALID_ARG; }
    const LdpcCode& code = codes_[fec];
    const int n = code.n, k = code.k, ncheck = n - k;
    const int frame_bytes = cap * bps / 8;
    const int ncws = frame_bytes * 8 > n ? 1 + (frame_bytes * 8) / n : 1; // compute_tb_len
    const int tb_payload_max = ncws * k / 8 - 4;
    if (tb_payload_max <= 0) { set_last_error("Misconfiguration: TB should be able to carry at least 1 user data bit"); return DTL_ERR_INVALID_ARG; }
    long offset = 0, used = 0, payload = 0, wpos = 0;
    int frame_no = first_frame_no, tb_count = first_tb_no, tb_bits = 0;
    int cur_seg_first = 0;
    auto emit = [&](long payload_bytes) {
        TxFrameInfo fi{};
        fi.payload_len = int(payload_bytes); fi.frame_no = frame_no & 0xFFF; fi.cnst = cnst; fi.fb_cnst = 0;
        fi.tb_no = tb_count & 0xFF; fi.fec_fb = 0; fi.tb_offset = int(offset) & 0xFFF; fi.fec = fec; fi.tb_payload = tb_bits & 0xFFFF;
        frames.push_back(fi);
        tframes_.push_back({ cur_seg_first, int(segs_.size()) - cur_seg_first, cnst, 0 });
        cur_seg_first = int(segs_.size());
        offset = used = payload = wpos = 0;
        frame_no = (frame_no + 1) & 0xFFF;
    };
    int64_t read = 0;
    while (read < n_user) {
        const uint32_t len = uint32_t(std::min<int64_t>(tb_payload_max, n_user - read));
        const long tb_len = long(ncws) * ncheck + long(len + 4) * 8; // [check | data] per codeword
        TxTb tb{ uint32_t(read), len, uint32_t(coded_total_), uint32_t(tb_len), ncws, fec };
        if (coded_total_ + uint64_t(tb_len) > 0xFFFFFFF0ull) { set_last_error("tx: coded stream exceeds 32-bit offsets; split the batch"); return DTL_ERR_CAPACITY; }
        tbs_.push_back(tb);
        coded_total_ += uint64_t(tb_len);
        tb_bits = int(len + 4) * 8;
        read += len;
        tb_count += 1;
        int used_frames = 0;
        long idx = 0;
        while (idx < tb_len) {
            const long remaining = tb_len - idx;
            const long bytes_left = (remaining + 7) / 8;
            const long avail = (cap - used) * bps / 8;
            const long out_bytes = std::min(bytes_left, avail);
            const long nbits = std::min(out_bytes * 8, remaining);
            if (wpos > cap) { set_last_error("tx: frame write position past the frame (reference would overrun)"); return DTL_ERR_CAPACITY; }
            segs_.push_back({ uint32_t(tb.coded_off + idx), uint32_t(nbits), int32_t(wpos), 0 });
            idx += nbits;
            if (out_bytes == avail) {
                emit(used * bps / 8 + out_bytes);
            } else if (offset == 0) {
                offset = out_bytes;
                payload += out_bytes;
                used = (out_bytes * 8 + bps - 1) / bps;
                wpos += used;
            } else {
                used = offset + (out_bytes * 8 + bps - 1) / bps; // sic (:459-460)
                payload += out_bytes;
                wpos += used;
            }
            used_frames++;
        }
        if ((used_frames == 1 || read >= n_user) && (payload > 0 || used > 0)) emit(payload);
    }
    return DTL_OK;
}

int Transmitter::process(const uint8_t* d_user, int64_t n_user, int cnst, int fec, int first_frame_no, int first_tb_no,
                         int gap, bool want_syms, hipStream_t st)
{
    if (n_user < 0 || (!d_user && n_user > 0) || gap < 0) { set_last_error("tx: bad input"); return DTL_ERR_INVALID_ARG; }
    if (n_user > int64_t(0xFFFFFFF0u)) { set_last_error("tx: at most 2^32 user bytes per call"); return DTL_ERR_INVALID_ARG; }
    DeviceScope dev;
    DTL_TRY(dev.enter(cfg.device));
    if (!uploaded_) DTL_TRY(upload_tables());
    DTL_TRY(plan(n_user, cnst, fec, first_frame_no, first_tb_no));
    const int n_frames = int(frames.size());
    n_tbs = int64_t(tbs_.size());
    const int64_t stride = int64_t(frame_samples()) + gap;
    n_samples = stride * n_frames;
    if (!n_frames) return DTL_OK;
    const int hb = hdr_bits();
    hdr_bits_.resize(size_t(n_frames) * hb);
    for (int f = 0; f < n_frames; f++) format_header_bits(frames[f], cfg.has_fec, hb, &hdr_bits_[size_t(f) * hb]);
    DTL_TRY(upload_vec(h_tbs_, d_tbs_, tbs_, st));
    DTL_TRY(upload_vec(h_segs_, d_segs_, segs_, st));
    DTL_TRY(upload_vec(h_tframes_, d_tframes_, tframes_, st));
    DTL_TRY(upload_vec(h_hdr_, d_hdr_bits, hdr_bits_, st));
    DTL_TRY(d_coded_.ensure(size_t(coded_total_) + 64));
    DTL_TRY(d_iq.ensure(size_t(n_samples) * sizeof(cf32)));
    uint8_t* syms = nullptr;
    if (want_syms) { DTL_TRY(d_syms.ensure(size_t(n_frames) * capacity_)); syms = d_syms.as<uint8_t>(); }
    if (gap) DTL_HIP_CHECK(hipMemsetAsync(d_iq.p, 0, size_t(n_samples) * sizeof(cf32), st));
    int max_len = 0, max_cws = 0;
    for (const auto& tb : tbs_) { max_len = std::max<int>(max_len, int(tb.user_len)); max_cws = std::max(max_cws, tb.ncws); }
    DTL_TRY(launch_tx_encode(code_tab_.data(), int(code_tab_.size()), d_tbs_.as<TxTb>(), int(tbs_.size()), max_len, max_cws,
                             d_user, d_coded_.as<uint8_t>(), st));
    DTL_TRY(launch_tx_frames(tab_, d_tframes_.as<TxFrame>(), n_frames, d_segs_.as<TxSeg>(), d_coded_.as<uint8_t>(),
                             d_hdr_bits.as<uint8_t>(), d_iq.as<cf32>(), stride, syms, st));
    DTL_HIP_CHECK(hipStreamSynchronize(st)); // the pinned staging buffers are reused by the next call
    return DTL_OK;
}

} // namespace dtl
