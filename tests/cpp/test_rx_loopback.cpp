// C++ host program over the C ABI alone (include/dtl_rx.h + the HIP runtime, no Python): build an
// ofdm_adaptive_config equivalent, generate frames with dtl_tx_*, decode them with dtl_rx_process and
// again in three arbitrary slices with dtl_rx_process_stream, and through dtl_rx_submit / dtl_rx_wait; the user bytes must come back.
// usage: test_rx_loopback <alist file> <constellation 1..4> <user bytes>
#include <dtl_rx.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#define CHECK_HIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::fprintf(stderr, "HIP: %s\n", hipGetErrorString(_e)); return 2; } } while (0)
#define CHECK_DTL(e) do { int _r = (e); if (_r != DTL_OK) { std::fprintf(stderr, "dtl: %d %s\n", _r, dtl_last_error()); return 3; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: %s alist cnst nbytes\n", argv[0]); return 1; }
    std::ifstream f(argv[1]);
    std::stringstream ss; ss << f.rdbuf();
    const std::string alist = ss.str();
    const int cnst = std::atoi(argv[2]);
    const int64_t n_user = std::atoll(argv[3]);

    // python/dtl/ofdm_adaptive_config.py:10-63, the 64-carrier defaults
    const int N = 64, cp = 16, L = 20;
    std::vector<int32_t> occ;
    for (int c = -26; c <= 26; c++) if (c != 0 && c != -21 && c != -7 && c != 7 && c != 21) occ.push_back(c);
    const int32_t occ_len = int32_t(occ.size());
    const int32_t pilots[4] = { -21, -7, 7, 21 };
    const float pilot_syms[8] = { 1, 0, 1, 0, 1, 0, -1, 0 };
    std::vector<float> sw1(2 * N, 0.0f), sw2(2 * N, 0.0f);
    uint32_t lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 16) & 1u; };
    for (int k = 0; k < N; k++) {             // shifted order: carrier c = k - N/2
        const int c = k - N / 2;
        if (c < -26 || c > 26 || c == 0) continue;
        // sync word 1 on the odd carriers only: the two halves of the symbol are negatives of each other, which is
        // the sign the Schmidl-Cox branch of the reference graph expects (its correlator taps are -1)
        if (c & 1) sw1[2 * k] = rnd() ? std::sqrt(2.0f) : -std::sqrt(2.0f);
        sw2[2 * k] = rnd() ? 1.0f : -1.0f;
    }
    const char* texts[1] = { alist.c_str() };
    const size_t lens[1] = { alist.size() };
    const double mcs_snr[4] = { -1e5, 11, 16, 21 };
    const int32_t mcs_cnst[4] = { 1, 2, 3, 4 }, mcs_fec[4] = { 1, 1, 1, 1 };
    dtl_rx_cfg cfg{};
    cfg.fft_len = N; cfg.cp_len = cp; cfg.frame_length = L; cfg.has_fec = 1;
    cfg.n_occ_sets = 1; cfg.max_occ = occ_len; cfg.occ_len = &occ_len; cfg.occ = occ.data();
    cfg.n_pilot_sets = 1; cfg.n_pilots = 4; cfg.pilot_carriers = pilots;
    cfg.n_pilot_sym_sets = 1; cfg.pilot_symbols = pilot_syms;
    cfg.sync_word1 = sw1.data(); cfg.sync_word2 = sw2.data();
    cfg.sync_threshold = 0.95f; cfg.eq_alpha = 0.1f; cfg.use_sync_correct = 1;   // frame_detect_bb on: the receiver default
    cfg.max_iterations = 15; cfg.n_fec_codes = 1; cfg.alist_texts = texts; cfg.alist_lens = lens;
    cfg.n_mcs = 4; cfg.mcs_snr = mcs_snr; cfg.mcs_cnst = mcs_cnst; cfg.mcs_fec = mcs_fec;
    cfg.initial_mcs = 1; cfg.decision_th = 5; cfg.hysteresis = 1.0;

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
    // frames back to back, as the reference's transmitter sends them (qa_ofdm_adaptive_txrx.py)
    CHECK_DTL(dtl_tx_process(tx, d_user, n_user, cnst, 1, 0, 0, /*gap=*/0, 0, nullptr, &t));

    // 100 zeros in front (qa_ofdm_adaptive_txrx.py:82-83), one frame of zeros behind
    const int64_t n = 100 + t.n_samples + t.frame_samples;
    float* d_x = nullptr;
    CHECK_HIP(hipMalloc(&d_x, size_t(n) * 8));
    CHECK_HIP(hipMemset(d_x, 0, size_t(n) * 8));
    CHECK_HIP(hipMemcpy(d_x + 2 * 100, t.d_iq, size_t(t.n_samples) * 8, hipMemcpyDeviceToDevice));

    dtl_rx_result r{};
    CHECK_DTL(dtl_rx_process(rx, d_x, n, 0, nullptr, &r));
    std::vector<uint8_t> got(size_t(r.n_payload_bytes));
    if (!got.empty()) CHECK_HIP(hipMemcpy(got.data(), r.d_payload, got.size(), hipMemcpyDeviceToHost));
    if (got != user) { std::fprintf(stderr, "batch: %zu bytes back, %zu sent\n", got.size(), user.size()); return 4; }
    const long long frames = (long long)r.n_payload_frames, tbs = (long long)r.n_tbs;

    // the same stream in three slices
    const int64_t cut[4] = { 0, n / 3 + 7, 2 * n / 3 - 11, n };
    std::vector<uint8_t> got2;
    for (int i = 0; i < 3; i++) {
        CHECK_DTL(dtl_rx_process_stream(rx, d_x + 2 * cut[i], cut[i + 1] - cut[i], 0, i == 2, nullptr, &r));
        std::vector<uint8_t> part(size_t(r.n_payload_bytes));
        if (!part.empty()) CHECK_HIP(hipMemcpy(part.data(), r.d_payload, part.size(), hipMemcpyDeviceToHost));
        got2.insert(got2.end(), part.begin(), part.end());
    }
    if (got2 != user) { std::fprintf(stderr, "stream: %zu bytes back, %zu sent\n", got2.size(), user.size()); return 5; }

    // batches handed over ahead: four submits through the handle's two lanes, two always in flight, results in order
    // (the whole stream, its first half, the whole stream, nothing)
    {
        const int64_t len[4] = { n, n / 2, n, 0 };
        size_t bytes[4] = { 0, 0, 0, 0 };
        CHECK_DTL(dtl_rx_submit(rx, d_x, len[0], 0, nullptr));
        for (int i = 0; i < 4; i++) {
            if (i + 1 < 4) CHECK_DTL(dtl_rx_submit(rx, d_x, len[i + 1], 0, nullptr));
            CHECK_DTL(dtl_rx_wait(rx, &r));
            bytes[i] = size_t(r.n_payload_bytes);
            if (i == 0 || i == 2) {
                std::vector<uint8_t> g(bytes[i]);
                if (!g.empty()) CHECK_HIP(hipMemcpy(g.data(), r.d_payload, g.size(), hipMemcpyDeviceToHost));
                if (g != user) { std::fprintf(stderr, "submit/wait %d: %zu bytes back, %zu sent\n", i, g.size(), user.size()); return 6; }
            }
        }
        if (!(bytes[1] > 0 && bytes[1] < bytes[0] && bytes[3] == 0)) { std::fprintf(stderr, "submit/wait: %zu %zu %zu %zu\n", bytes[0], bytes[1], bytes[2], bytes[3]); return 7; }
        if (dtl_rx_wait(rx, &r) != DTL_ERR_INVALID_ARG) { std::fprintf(stderr, "wait on an empty queue must be refused\n"); return 8; }
    }
    std::printf("OK frames=%lld tbs=%lld bytes=%zu\n", frames, tbs, got.size());
    dtl_rx_destroy(rx);
    dtl_tx_destroy(tx);
    (void)hipFree(d_user); (void)hipFree(d_x);
    return 0;
}
