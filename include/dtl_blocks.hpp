// dtl_blocks.hpp -- C++ mirrors of the two gr-dtl leaf blocks of the receive path that carry arithmetic, on top of
// the C ABI (dtl_rx.h), GNU-Radio-free: same class names, factory names and argument lists as the reference, so a
// GR shim block (shim/) only has to forward its constructor arguments and its work() buffers.
//
//   gr::dtl::ofdm_adaptive_payload_equalizer::make     include/gnuradio/dtl/ofdm_adaptive_equalizer.h:106-117
//   gr::dtl::ofdm_adaptive_feedback_decision::make     include/gnuradio/dtl/ofdm_adaptive_feedback_decision.h:49-52
//   gr::dtl::ofdm_adaptive_frame_equalizer_vcvc::make  include/gnuradio/dtl/ofdm_adaptive_frame_equalizer_vcvc.h:30-36
//   gr::dtl::ofdm_adaptive_fec_decoder::make           include/gnuradio/dtl/ofdm_adaptive_fec_decoder.h:27
//
// What replaces work(): the reference blocks are handed ONE frame per call by the scheduler and read the frame's
// parameters from stream tags; these objects take a BATCH of frames (device vectors) with the tag values as plain
// structs and return what the blocks emit as tags / messages as plain structs.  Exceptions of the reference
// (std::invalid_argument("Missing constellation tag."), std::runtime_error("Tags missing"), ...) are thrown from
// the same conditions.
#pragma once

#include "dtl_fec.hpp"

#include <complex>
#include <cstdint>
#include <cstring>
#include <utility>

namespace gr {
namespace dtl {

#ifndef INCLUDED_GR_COMPLEX_H
typedef std::complex<float> gr_complex;
#endif

// include/gnuradio/dtl/ofdm_adaptive_utils.h:22-28
enum class constellation_type_t { UNKNOWN = 0, BPSK, QPSK, PSK8, QAM16 };

// include/gnuradio/dtl/ofdm_adaptive_frame_snr.h:27-42 -- the estimator is fixed (mpsk_snr_est_simple), only its
// constructor argument is kept
class ofdm_adaptive_frame_snr_base
{
public:
    virtual ~ofdm_adaptive_frame_snr_base() = default;
};
class ofdm_adaptive_frame_snr_simple : public ofdm_adaptive_frame_snr_base
{
public:
    explicit ofdm_adaptive_frame_snr_simple(double alpha) : d_alpha(alpha) {}
    double d_alpha;
};

class ofdm_adaptive_equalizer_base
{
public:
    typedef std::shared_ptr<ofdm_adaptive_equalizer_base> sptr;
    virtual ~ofdm_adaptive_equalizer_base() = default;
    int d_fft_len = 0, d_symbols_skipped = 0;
    float d_alpha = 0.1f;
    bool d_input_is_shifted = true;
    std::vector<std::vector<int>> d_occupied_carriers, d_pilot_carriers;
    std::vector<std::vector<gr_complex>> d_pilot_symbols;
    std::vector<constellation_type_t> d_constellations;
};

class ofdm_adaptive_payload_equalizer : public ofdm_adaptive_equalizer_base
{
public:
    static sptr make(int fft_len, const std::vector<constellation_type_t>& constellations,
                     const std::shared_ptr<ofdm_adaptive_frame_snr_base> snr_est,
                     const std::vector<std::vector<int>>& occupied_carriers = std::vector<std::vector<int>>(),
                     const std::vector<std::vector<int>>& pilot_carriers = std::vector<std::vector<int>>(),
                     const std::vector<std::vector<gr_complex>>& pilot_symbols = std::vector<std::vector<gr_complex>>(),
                     int symbols_skipped = 0, float alpha = 0.1, bool input_is_shifted = true)
    {
        (void)snr_est;
        if (!input_is_shifted) throw std::invalid_argument("ofdm_adaptive_payload_equalizer: only shifted input is supported");
        auto e = std::make_shared<ofdm_adaptive_payload_equalizer>();
        e->d_fft_len = fft_len; e->d_constellations = constellations; e->d_occupied_carriers = occupied_carriers;
        e->d_pilot_carriers = pilot_carriers; e->d_pilot_symbols = pilot_symbols; e->d_symbols_skipped = symbols_skipped;
        e->d_alpha = alpha;
        return e;
    }
};

// include/gnuradio/dtl/ofdm_adaptive_feedback_decision.h:20-21; a LUT row is (snr threshold, (constellation, FEC
// scheme index)) -- python/dtl/ofdm_receiver.py:164-167
typedef std::pair<constellation_type_t, unsigned char> ofdm_adaptive_feedback_t;
typedef int mcs_id_t;
typedef std::pair<double, ofdm_adaptive_feedback_t> mcs_entry_t;

class ofdm_adaptive_feedback_decision_base
{
public:
    typedef std::shared_ptr<ofdm_adaptive_feedback_decision_base> sptr;
    virtual ~ofdm_adaptive_feedback_decision_base() = default;
    double d_hysterisis = 1;
    int d_decision_th = 5, d_initial_mcs_id = 0;
    std::vector<mcs_entry_t> d_lut;
};
class ofdm_adaptive_feedback_decision : public ofdm_adaptive_feedback_decision_base
{
public:
    static sptr make(double hysterisis, int decision_th, const std::vector<std::pair<double, ofdm_adaptive_feedback_t>>& lut,
                     mcs_id_t initial_mcs_id = 0)
    {
        auto d = std::make_shared<ofdm_adaptive_feedback_decision>();
        d->d_hysterisis = hysterisis; d->d_decision_th = decision_th; d->d_lut = lut; d->d_initial_mcs_id = initial_mcs_id;
        return d;
    }
};

// what the frame equaliser reads from the tags of one frame / what it publishes for it
struct frame_eq_tags { constellation_type_t constellation; int carr_offset; int packet_len; /* QAM symbols */ };
struct frame_eq_result {
    double snr_db, noise;                       // estimated_snr_tag_key / noise_tag_key
    constellation_type_t feedback_constellation; // feedback_constellation_key
    unsigned char feedback_fec;                  // feedback_fec_key
};

class ofdm_adaptive_frame_equalizer_vcvc
{
    dtl_rx* d_h = nullptr;
    int d_fft_len = 0, d_frame_len = 0;
    std::vector<double> d_snr_th;
    std::vector<int32_t> d_cnst, d_fec;
    double d_hyst = 1;
    int d_th = 5, d_initial = 0;

public:
    typedef std::shared_ptr<ofdm_adaptive_frame_equalizer_vcvc> sptr;
    ~ofdm_adaptive_frame_equalizer_vcvc() { dtl_rx_destroy(d_h); }
    // `frame_length` (OFDM symbols per frame) is what the reference block learns from the tagged-stream length tag
    // at run time; the batched form needs it up front.
    static sptr make(ofdm_adaptive_equalizer_base::sptr equalizer, ofdm_adaptive_feedback_decision_base::sptr feedback_decision,
                     int cp_len, const std::string& len_tag_key = "len_key", const std::string& frame_no_key = "frame_no_key",
                     bool propagate_channel_state = false, bool propagate_feedback_tags = false, int frame_length = 20)
    {
        (void)len_tag_key; (void)frame_no_key; (void)propagate_channel_state; (void)propagate_feedback_tags;
        if (!equalizer || !feedback_decision) throw std::invalid_argument("ofdm_adaptive_frame_equalizer_vcvc: null argument");
        auto b = std::make_shared<ofdm_adaptive_frame_equalizer_vcvc>();
        const int N = equalizer->d_fft_len;
        b->d_fft_len = N; b->d_frame_len = frame_length;
        // flatten the carrier tables the way dtl_rx_cfg wants them
        const auto& occ = equalizer->d_occupied_carriers;
        const auto& pil = equalizer->d_pilot_carriers;
        const auto& psym = equalizer->d_pilot_symbols;
        if (occ.empty() || pil.empty() || psym.empty()) throw std::invalid_argument("ofdm_adaptive_frame_equalizer_vcvc: carrier tables missing");
        std::vector<int32_t> occ_len, occ_flat, pil_flat;
        size_t max_occ = 0;
        for (auto& o : occ) max_occ = std::max(max_occ, o.size());
        for (auto& o : occ) {
            occ_len.push_back(int32_t(o.size()));
            for (size_t i = 0; i < max_occ; i++) occ_flat.push_back(i < o.size() ? o[i] : 0);
        }
        // pilot sets may differ in size: every set is padded to the largest one with DTL_NO_CARRIER, every pilot-symbol
        // row with 0 (dtl_rx_cfg reads n_pilot_sets x n_pilots entries)
        size_t n_pil = 1;
        for (auto& p : pil) n_pil = std::max(n_pil, p.size());
        for (auto& row : psym) n_pil = std::max(n_pil, row.size());
        for (auto& p : pil)
            for (size_t i = 0; i < n_pil; i++) pil_flat.push_back(i < p.size() ? int32_t(p[i]) : int32_t(DTL_NO_CARRIER));
        std::vector<float> psym_flat;
        for (auto& row : psym)
            for (size_t i = 0; i < n_pil; i++) {
                psym_flat.push_back(i < row.size() ? row[i].real() : 0.0f);
                psym_flat.push_back(i < row.size() ? row[i].imag() : 0.0f);
            }
        // the equaliser block does not look at the synchronisation words; the handle wants well-formed ones
        std::vector<float> sw1(2 * size_t(N), 0.0f), sw2(2 * size_t(N), 0.0f);
        for (int c : occ[0]) {
            const int k = ((c % N) + N + N / 2) % N;
            sw2[2 * size_t(k)] = 1.0f;
            if (c & 1) sw1[2 * size_t(k)] = 1.41421356f;
        }
        for (auto& e : feedback_decision->d_lut) {
            b->d_snr_th.push_back(e.first);
            b->d_cnst.push_back(int32_t(e.second.first));
            b->d_fec.push_back(int32_t(e.second.second));
        }
        b->d_hyst = feedback_decision->d_hysterisis; b->d_th = feedback_decision->d_decision_th;
        b->d_initial = feedback_decision->d_initial_mcs_id;
        dtl_rx_cfg cfg{};
        cfg.fft_len = N; cfg.cp_len = cp_len; cfg.frame_length = frame_length;
        cfg.has_fec = equalizer->d_symbols_skipped == 2;   // header_len = symbols_skipped (ofdm_receiver.py:66-68,162)
        cfg.n_occ_sets = int32_t(occ.size()); cfg.max_occ = int32_t(max_occ); cfg.occ_len = occ_len.data(); cfg.occ = occ_flat.data();
        cfg.n_pilot_sets = int32_t(pil.size()); cfg.n_pilots = int32_t(n_pil); cfg.pilot_carriers = pil_flat.data();
        cfg.n_pilot_sym_sets = int32_t(psym.size()); cfg.pilot_symbols = psym_flat.data();
        cfg.sync_word1 = sw1.data(); cfg.sync_word2 = sw2.data();
        cfg.sync_threshold = 0.95f; cfg.eq_alpha = equalizer->d_alpha; cfg.max_iterations = 15;
        cfg.n_mcs = int32_t(b->d_snr_th.size()); cfg.mcs_snr = b->d_snr_th.data(); cfg.mcs_cnst = b->d_cnst.data(); cfg.mcs_fec = b->d_fec.data();
        cfg.initial_mcs = b->d_initial; cfg.decision_th = b->d_th; cfg.hysteresis = b->d_hyst;
        detail::check(dtl_rx_create(&cfg, &b->d_h));
        return b;
    }

    // n frames: d_frames n x frame_length x fft_len (device, fft_vcc output), d_taps n x fft_len (the
    // "ofdm_sync_chan_taps" tag values).  d_out / d_out_soft: out0 / out1 of the block (device, may be null);
    // d_serialized: the ofdm_serializer_vcc behind out1, n x frame_capacity (may be null).
    void work(const gr_complex* d_frames, const gr_complex* d_taps, const std::vector<frame_eq_tags>& tags, gr_complex* d_out,
              gr_complex* d_out_soft, gr_complex* d_serialized, std::vector<frame_eq_result>& results, hipStream_t stream = nullptr)
    {
        const size_t n = tags.size();
        std::vector<dtl_frame_eq_in> in(n);
        for (size_t i = 0; i < n; i++) {
            if (tags[i].constellation == constellation_type_t::UNKNOWN) throw std::invalid_argument("Missing constellation tag.");
            in[i] = { int32_t(tags[i].constellation), tags[i].carr_offset, tags[i].packet_len, 0 };
        }
        std::vector<dtl_frame_eq_out> out(n);
        detail::check(dtl_equalize_frames(d_h, reinterpret_cast<const float*>(d_frames), reinterpret_cast<const float*>(d_taps), in.data(),
                                          int64_t(n), reinterpret_cast<float*>(d_out), reinterpret_cast<float*>(d_out_soft),
                                          reinterpret_cast<float*>(d_serialized), out.data(), stream));
        // ofdm_adaptive_feedback_decision::get_feedback over the frames' SNR estimates (state kept across calls
        // would need a handle of its own; a batch starts from the initial MCS like a fresh block)
        std::vector<double> snr(n);
        for (size_t i = 0; i < n; i++) snr[i] = out[i].snr_db;
        std::vector<int32_t> fc(n), ff(n);
        if (n) detail::check(dtl_feedback_sequence(int32_t(d_snr_th.size()), d_snr_th.data(), d_cnst.data(), d_fec.data(), d_hyst, d_th,
                                                  d_initial, snr.data(), int64_t(n), fc.data(), ff.data()));
        results.resize(n);
        for (size_t i = 0; i < n; i++) results[i] = { out[i].snr_db, out[i].noise, constellation_type_t(fc[i]), (unsigned char)ff[i] };
    }
    int frame_length() const { return d_frame_len; }
};

// one decoded transport block: the monitor message of the block (fec_decoder_impl.cc:183-195) + its bytes
struct tb_result {
    int tb_no, tb_payload_bits, avg_it;
    bool crc_ok;
    std::vector<unsigned char> user_bytes;   // without the CRC-32; empty when the CRC failed
};

class ofdm_adaptive_fec_decoder
{
    std::vector<fec_dec::sptr> d_decoders;
    std::vector<int32_t> d_n, d_k;
    int d_frame_capacity = 0, d_max_bps = 0;

    static uint32_t crc32(const unsigned char* p, size_t len) // crc_util.cc:23: crc(32, 0x04C11DB7, ~0, ~0, refin, refout)
    {
        uint32_t c = 0xFFFFFFFFu;
        for (size_t i = 0; i < len; i++) {
            c ^= p[i];
            for (int b = 0; b < 8; b++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        }
        return ~c;
    }

public:
    typedef std::shared_ptr<ofdm_adaptive_fec_decoder> sptr;
    static sptr make(const std::vector<fec_dec::sptr>& decoders, int frame_capacity, int max_bps, const std::string& len_key)
    {
        (void)len_key;
        auto b = std::make_shared<ofdm_adaptive_fec_decoder>();
        b->d_decoders = decoders; b->d_frame_capacity = frame_capacity; b->d_max_bps = max_bps;
        for (size_t i = 1; i < decoders.size(); i++) {   // index 0 is nullptr ("no FEC")
            if (!decoders[i]) throw std::invalid_argument("ofdm_adaptive_fec_decoder: null decoder");
            b->d_n.push_back(decoders[i]->get_n());
            b->d_k.push_back(decoders[i]->get_k());
        }
        return b;
    }

    // frames: the header tags of a run of frames + where each frame's soft bits start in d_llr (llr_offset);
    // d_llr: device soft-bit stream (the soft demapper's output).  -> every transport block the run completes.
    std::vector<tb_result> work(const std::vector<dtl_rx_frame_info>& frames, const float* d_llr, hipStream_t stream = nullptr)
    {
        std::vector<tb_result> out;
        if (frames.empty()) return out;
        for (auto& f : frames)
            if (f.header_ok && f.llr_offset >= 0 && (f.cnst < 1 || f.cnst > 4)) throw std::runtime_error("Tags missing");
        int64_t n_cw = 0, n_sp = 0, cw_cap = 16, sp_cap = int64_t(2 * frames.size() + 4);
        for (auto& f : frames) cw_cap += int64_t(f.n_payload_syms) * 4 / 50 + 4;
        dtl_cw_desc* d_desc = nullptr;
        detail::hip_check(hipMalloc(reinterpret_cast<void**>(&d_desc), size_t(cw_cap) * sizeof(dtl_cw_desc)));
        std::vector<dtl_tb_span> spans(static_cast<size_t>(sp_cap));
        int rc = dtl_tb_gather(d_n.data(), d_k.data(), int32_t(d_n.size()), d_frame_capacity, d_max_bps, frames.data(),
                               int64_t(frames.size()), d_desc, cw_cap, &n_cw, spans.data(), sp_cap, &n_sp, stream);
        if (rc != DTL_OK) { (void)hipFree(d_desc); detail::check(rc); }
        spans.resize(size_t(n_sp));
        if (spans.empty()) { (void)hipFree(d_desc); return out; }
        const size_t n_bits = size_t(spans.back().bits_off) + size_t(spans.back().tb_payload_bits) + 64;
        unsigned char* d_bits = nullptr;
        int32_t* d_nit = nullptr;
        detail::hip_check(hipMalloc(reinterpret_cast<void**>(&d_bits), n_bits));
        detail::hip_check(hipMalloc(reinterpret_cast<void**>(&d_nit), size_t(n_cw) * sizeof(int32_t)));
        // one launch per run of blocks that use the same code
        for (size_t s = 0; s < spans.size();) {
            size_t e = s;
            int64_t cws = 0;
            while (e < spans.size() && spans[e].fec == spans[s].fec) cws += spans[e++].n_cw;
            auto* dec = dynamic_cast<ldpc_dec*>(d_decoders.at(size_t(spans[s].fec)).get());
            if (!dec) throw std::runtime_error("ofdm_adaptive_fec_decoder: decoder is not a device decoder");
            detail::check(dtl_ldpc_decode_gather(dec->handle(), d_llr, d_desc + spans[s].cw_first, size_t(cws), d_bits,
                                                 d_nit + spans[s].cw_first, stream));
            s = e;
        }
        std::vector<unsigned char> bits(n_bits);
        std::vector<int32_t> nit(static_cast<size_t>(n_cw));
        detail::hip_check(hipMemcpyAsync(bits.data(), d_bits, n_bits, hipMemcpyDeviceToHost, stream));
        detail::hip_check(hipMemcpyAsync(nit.data(), d_nit, nit.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        detail::hip_check(hipStreamSynchronize(stream));
        (void)hipFree(d_desc); (void)hipFree(d_bits); (void)hipFree(d_nit);
        for (auto& sp : spans) {
            tb_result r{ sp.tb_no, sp.tb_payload_bits, 0, false, {} };
            long it = 0;
            for (int c = 0; c < sp.n_cw; c++) it += nit[size_t(sp.cw_first) + size_t(c)];
            r.avg_it = sp.n_cw ? int(it / sp.n_cw) : 0;                       // tb_decoder.cc:191
            std::vector<unsigned char> bytes(size_t(sp.tb_payload_bits / 8));
            for (size_t b = 0; b < bytes.size(); b++) {                        // repack 1 -> 8, LSB first
                unsigned v = 0;
                for (int j = 0; j < 8; j++) v |= unsigned(bits[size_t(sp.bits_off) + 8 * b + size_t(j)] & 1u) << j;
                bytes[b] = (unsigned char)v;
            }
            if (bytes.size() >= 4) {
                const uint32_t c = crc32(bytes.data(), bytes.size() - 4);
                uint32_t got;
                std::memcpy(&got, bytes.data() + bytes.size() - 4, 4);
                r.crc_ok = (c == got);
            }
            if (r.crc_ok) r.user_bytes.assign(bytes.begin(), bytes.end() - 4);
            out.push_back(std::move(r));
        }
        return out;
    }
};

} // namespace dtl
} // namespace gr
