// include/dtl_blocks.hpp driven the way a GNU Radio shim would drive it: the C++ mirrors of
// gr::dtl::ofdm_adaptive_frame_equalizer_vcvc and gr::dtl::ofdm_adaptive_fec_decoder are built with the reference's
// factory argument lists and fed the frequency-domain frames / soft bits of a noise-free TX -> RX run; their outputs
// must equal what the whole-receiver call (dtl_rx_process) produced for the same frames.
// usage: test_blocks_shim <alist file> <constellation 1..4> <user bytes>
#include <dtl_blocks.hpp>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#define CHECK_HIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::fprintf(stderr, "HIP: %s\n", hipGetErrorString(_e)); return 2; } } while (0)
#define CHECK_DTL(e) do { int _r = (e); if (_r != DTL_OK) { std::fprintf(stderr, "dtl: %d %s\n", _r, dtl_last_error()); return 3; } } while (0)

using namespace gr::dtl;

int main(int argc, char** argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: %s alist cnst nbytes\n", argv[0]); return 1; }
    std::ifstream f(argv[1]);
    std::stringstream ss; ss << f.rdbuf();
    const std::string alist = ss.str();
    const int cnst = std::atoi(argv[2]);
    const int64_t n_user = std::atoll(argv[3]);
    const bool compile_only = argc > 4;

    const int N = 64, cp = 16, L = 20, h = 2, P = N + cp;
    std::vector<int> occ;
    for (int c = -26; c <= 26; c++) if (c != 0 && c != -21 && c != -7 && c != 7 && c != 21) occ.push_back(c);
    std::vector<int32_t> occ32(occ.begin(), occ.end());
    const int32_t occ_len = int32_t(occ.size());
    const int32_t pilots[4] = { -21, -7, 7, 21 };
    const float pilot_syms[8] = { 1, 0, 1, 0, 1, 0, -1, 0 };
    std::vector<float> sw1(2 * N, 0.0f), sw2(2 * N, 0.0f);
    uint32_t lcg = 4242u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 16) & 1u; };
    for (int k = 0; k < N; k++) {
        const int c = k - N / 2;
        if (c < -26 || c > 26 || c == 0) continue;
        if (c & 1) sw1[2 * k] = rnd() ? std::sqrt(2.0f) : -std::sqrt(2.0f);
        sw2[2 * k] = rnd() ? 1.0f : -1.0f;
    }
    const char* texts[1] = { alist.c_str() };
    const size_t lens[1] = { alist.size() };
    const double mcs_snr[4] = { -1e5, 11, 16, 21 };
    const int32_t mcs_cnst[4] = { 1, 2, 3, 4 }, mcs_fec[4] = { 1, 1, 1, 1 };
    dtl_rx_cfg cfg{};
    cfg.fft_len = N; cfg.cp_len = cp; cfg.frame_length = L; cfg.has_fec = 1;
    cfg.n_occ_sets = 1; cfg.max_occ = occ_len; cfg.occ_len = &occ_len; cfg.occ = occ32.data();
    cfg.n_pilot_sets = 1; cfg.n_pilots = 4; cfg.pilot_carriers = pilots;
    cfg.n_pilot_sym_sets = 1; cfg.pilot_symbols = pilot_syms;
    cfg.sync_word1 = sw1.data(); cfg.sync_word2 = sw2.data();
    cfg.sync_threshold = 0.95f; cfg.eq_alpha = 0.1f; cfg.use_sync_correct = 1;
    cfg.max_iterations = 15; cfg.n_fec_codes = 1; cfg.alist_texts = texts; cfg.alist_lens = lens;
    cfg.n_mcs = 4; cfg.mcs_snr = mcs_snr; cfg.mcs_cnst = mcs_cnst; cfg.mcs_fec = mcs_fec;
    cfg.initial_mcs = 1; cfg.decision_th = 5; cfg.hysteresis = 1.0;

    // ---- the reference's constructor calls (python/dtl/ofdm_receiver.py:154-174, 215-221)
    auto equalizer = ofdm_adaptive_payload_equalizer::make(
        N, { constellation_type_t::BPSK, constellation_type_t::QPSK, constellation_type_t::PSK8, constellation_type_t::QAM16 },
        std::make_shared<ofdm_adaptive_frame_snr_simple>(0.1), { occ }, { { -21, -7, 7, 21 } },
        { { gr_complex(1, 0), gr_complex(1, 0), gr_complex(1, 0), gr_complex(-1, 0) } }, /*symbols_skipped=*/h, /*alpha=*/0.1f, true);
    std::vector<std::pair<double, ofdm_adaptive_feedback_t>> lut;
    for (int i = 0; i < 4; i++) lut.push_back({ mcs_snr[i], { constellation_type_t(mcs_cnst[i]), (unsigned char)mcs_fec[i] } });
    auto feedback = ofdm_adaptive_feedback_decision::make(1, 5, lut, 1);
    auto frame_eq = ofdm_adaptive_frame_equalizer_vcvc::make(equalizer, feedback, cp, "frame_length", "frame_no", true, true, L);
    {   // pilot sets of different sizes (and a shorter pilot-symbol row): the tables are padded to the largest set
        auto eq2 = ofdm_adaptive_payload_equalizer::make(
            N, { constellation_type_t::BPSK }, std::make_shared<ofdm_adaptive_frame_snr_simple>(0.1), { occ },
            { { -21, -7, 7, 21 }, { -21, 21 } },
            { { gr_complex(1, 0), gr_complex(1, 0), gr_complex(1, 0), gr_complex(-1, 0) }, { gr_complex(-1, 0), gr_complex(1, 0) } },
            h, 0.1f, true);
        auto fe2 = ofdm_adaptive_frame_equalizer_vcvc::make(eq2, feedback, cp, "frame_length", "frame_no", true, true, L);
        if (!fe2) { std::fprintf(stderr, "ragged pilot sets: no block\n"); return 3; }
    }
    auto decoders = make_ldpc_decoders({ std::string(argv[1]) });
    const int frame_capacity = L * int(occ.size());
    auto fec_decoder = ofdm_adaptive_fec_decoder::make(decoders, frame_capacity, 4, "packet_len");
    if (compile_only) { std::printf("OK built\n"); return 0; }

    // ---- TX -> whole-receiver call, keeping the per-stage outputs
    dtl_tx* tx = nullptr;
    dtl_rx* rx = nullptr;
    CHECK_DTL(dtl_tx_create(&cfg, &tx));
    CHECK_DTL(dtl_rx_create(&cfg, &rx));
    std::vector<uint8_t> user(n_user);
    for (auto& b : user) { lcg = lcg * 1664525u + 1013904223u; b = uint8_t(lcg >> 24); }
    uint8_t* d_user = nullptr;
    CHECK_HIP(hipMalloc(&d_user, size_t(n_user) + 1));
    CHECK_HIP(hipMemcpy(d_user, user.data(), size_t(n_user), hipMemcpyHostToDevice));
    dtl_tx_result t{};
    CHECK_DTL(dtl_tx_process(tx, d_user, n_user, cnst, 1, 0, 0, /*gap=*/0, 0, nullptr, &t));   // frames back to back
    const int64_t n = 100 + t.n_samples + t.frame_samples;
    float* d_x = nullptr;
    CHECK_HIP(hipMalloc(&d_x, size_t(n) * 8));
    CHECK_HIP(hipMemset(d_x, 0, size_t(n) * 8));
    CHECK_HIP(hipMemcpy(d_x + 2 * 100, t.d_iq, size_t(t.n_samples) * 8, hipMemcpyDeviceToDevice));
    dtl_rx_result r{};
    CHECK_DTL(dtl_rx_process(rx, d_x, n, DTL_RX_WANT_EQ_HARD | DTL_RX_WANT_EQ_SOFT | DTL_RX_WANT_SOFT_SYMS, nullptr, &r));
    std::vector<uint8_t> got(size_t(r.n_payload_bytes));
    if (!got.empty()) CHECK_HIP(hipMemcpy(got.data(), r.d_payload, got.size(), hipMemcpyDeviceToHost));
    if (got != user) { std::fprintf(stderr, "receiver: %zu bytes back, %zu sent\n", got.size(), user.size()); return 4; }
    std::vector<dtl_rx_frame_info> frames(r.frames, r.frames + r.n_frames);
    std::vector<dtl_rx_frame_info> pf;                                 // frames with a payload, in order
    for (auto& fi : frames) if (fi.payload_index >= 0) pf.push_back(fi);
    const size_t nf = pf.size();
    if (nf == 0 || int64_t(nf) != r.n_payload_frames) { std::fprintf(stderr, "no frames\n"); return 5; }

    // ---- the frames as fft_vcc hands them to the equaliser block: cut (delay N+cp, CP stripped) and transform.
    // Noise-free stream: the Schmidl-Cox angle is exactly zero, so the fine-frequency mixer is the identity.
    float *d_time = nullptr, *d_freq = nullptr, *d_taps = nullptr;
    CHECK_HIP(hipMalloc(&d_time, nf * L * N * 8));
    CHECK_HIP(hipMalloc(&d_freq, nf * L * N * 8));
    CHECK_HIP(hipMalloc(&d_taps, nf * N * 8));
    for (size_t i = 0; i < nf; i++) {
        for (int s = 0; s < L; s++) {
            const int64_t src = pf[i].trigger + int64_t(2 + h + s) * P + cp - P;
            CHECK_HIP(hipMemcpy(d_time + 2 * ((i * L + s) * N), d_x + 2 * src, size_t(N) * 8, hipMemcpyDeviceToDevice));
        }
        CHECK_HIP(hipMemcpy(d_taps + 2 * (i * N), r.d_taps + 2 * (size_t(pf[i].cand) * N), size_t(N) * 8, hipMemcpyDeviceToDevice));
    }
    CHECK_DTL(dtl_fft_batch(d_time, d_freq, int64_t(nf) * L, N, 1, nullptr));

    // ---- gr::dtl::ofdm_adaptive_frame_equalizer_vcvc
    std::vector<frame_eq_tags> tags(nf);
    for (size_t i = 0; i < nf; i++) tags[i] = { constellation_type_t(pf[i].cnst), pf[i].carr_offset, pf[i].n_payload_syms };
    gr_complex *d_hard = nullptr, *d_soft = nullptr, *d_ser = nullptr;
    CHECK_HIP(hipMalloc(&d_hard, nf * L * N * 8));
    CHECK_HIP(hipMalloc(&d_soft, nf * L * N * 8));
    CHECK_HIP(hipMalloc(&d_ser, nf * size_t(frame_capacity) * 8));
    std::vector<frame_eq_result> res;
    try {
        frame_eq->work(reinterpret_cast<const gr_complex*>(d_freq), reinterpret_cast<const gr_complex*>(d_taps), tags, d_hard, d_soft, d_ser, res);
    } catch (const std::exception& e) { std::fprintf(stderr, "frame equaliser: %s\n", e.what()); return 6; }
    auto same = [&](const void* d_a, const void* d_b, size_t floats, const char* what) -> bool {
        std::vector<float> a(floats), b(floats);
        if (hipMemcpy(a.data(), d_a, floats * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (hipMemcpy(b.data(), d_b, floats * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
        size_t bad = 0;
        for (size_t i = 0; i < floats; i++) if (!(a[i] == b[i])) bad++;
        if (bad) std::fprintf(stderr, "%s: %zu of %zu values differ\n", what, bad, floats);
        return bad == 0;
    };
    if (!same(d_hard, r.d_eq_hard, nf * L * N * 2, "eq_hard")) return 7;
    if (!same(d_soft, r.d_eq_soft, nf * L * N * 2, "eq_soft")) return 7;
    if (!same(d_ser, r.d_soft_syms, nf * size_t(frame_capacity) * 2, "serializer")) return 7;
    for (size_t i = 0; i < nf; i++) {
        if (!(res[i].snr_db == pf[i].snr_db) || !(res[i].noise == pf[i].noise)) { std::fprintf(stderr, "snr of frame %zu differs\n", i); return 8; }
        if (int(res[i].feedback_constellation) != pf[i].feedback_cnst || int(res[i].feedback_fec) != pf[i].feedback_fec) {
            std::fprintf(stderr, "feedback of frame %zu differs\n", i); return 8;
        }
    }
    // the reference's exception for a frame without a constellation tag
    bool threw = false;
    try {
        std::vector<frame_eq_tags> bad_tags(1, { constellation_type_t::UNKNOWN, 0, 10 });
        frame_eq->work(reinterpret_cast<const gr_complex*>(d_freq), reinterpret_cast<const gr_complex*>(d_taps), bad_tags, nullptr, nullptr, nullptr, res);
    } catch (const std::invalid_argument& e) { threw = std::string(e.what()) == "Missing constellation tag."; }
    if (!threw) { std::fprintf(stderr, "missing-tag exception not raised\n"); return 9; }

    // ---- gr::dtl::ofdm_adaptive_fec_decoder on the receiver's soft bits
    std::vector<tb_result> tbs;
    try { tbs = fec_decoder->work(frames, r.d_llr); } catch (const std::exception& e) { std::fprintf(stderr, "fec decoder: %s\n", e.what()); return 10; }
    if (int64_t(tbs.size()) != r.n_tbs) { std::fprintf(stderr, "%zu blocks, receiver had %lld\n", tbs.size(), (long long)r.n_tbs); return 11; }
    std::vector<uint8_t> got2;
    for (size_t i = 0; i < tbs.size(); i++) {
        if (tbs[i].tb_no != r.tbs[i].tb_no || int(tbs[i].crc_ok) != r.tbs[i].crc_ok || tbs[i].avg_it != r.tbs[i].avg_it) {
            std::fprintf(stderr, "block %zu differs\n", i); return 12;
        }
        got2.insert(got2.end(), tbs[i].user_bytes.begin(), tbs[i].user_bytes.end());
    }
    if (got2 != user) { std::fprintf(stderr, "fec decoder: %zu bytes back, %zu sent\n", got2.size(), user.size()); return 13; }
    std::printf("OK frames=%zu tbs=%zu bytes=%zu\n", nf, tbs.size(), got2.size());
    dtl_rx_destroy(rx);
    dtl_tx_destroy(tx);
    return 0;
}
