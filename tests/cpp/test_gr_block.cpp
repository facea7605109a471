// Executes shim/lib/ofdm_adaptive_rx_hip_impl.cc -- the gr::block a gr-dtl maintainer would compile -- against the
// functional GNU Radio stand-in of tests/cpp/mock_gnuradio/: gr::mock::harness plays the scheduler thread around
// general_work (caller-owned buffers with guard words, ragged ninput_items, small noutput_items, consume / produce
// bookkeeping, stored tags, captured messages).  The program reads a configuration and an IQ stream written by
// tests/test_gr_block_gpu.py, feeds the stream, and writes every port's items, every tag and every message to files;
// the Python test compares them with the oracle.
// usage: test_gr_block <cfg.txt> <iq.c64> <out dir> <async 0|1> <seed> <max noutput_items> [propagate 0|1]
#include <gnuradio/dtl_hip/ofdm_adaptive_rx_hip.h>

#include <hip/hip_runtime_api.h>

#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

static std::vector<std::string> split(const std::string& s)
{
    std::istringstream is(s);
    std::vector<std::string> v;
    std::string w;
    while (is >> w) v.push_back(w);
    return v;
}

static std::string show(const pmt::pmt_t& p)
{
    char buf[64];
    if (!p) return "null";
    switch (p->kind) {
    case pmt::pmt_base::SYMBOL: return "sym:" + p->sym;
    case pmt::pmt_base::LONG: return "long:" + std::to_string(p->l);
    case pmt::pmt_base::UINT64: return "u64:" + std::to_string(p->u);
    case pmt::pmt_base::DOUBLE: std::snprintf(buf, sizeof(buf), "double:%.17g", p->d); return buf;
    case pmt::pmt_base::BOOL: return p->b ? "bool:1" : "bool:0";
    case pmt::pmt_base::C32VECTOR: {
        std::string s = "c32:" + std::to_string(p->c32.size());
        for (auto& z : p->c32) { std::snprintf(buf, sizeof(buf), ",%.9g,%.9g", z.real(), z.imag()); s += buf; }
        return s;
    }
    case pmt::pmt_base::DICT: {
        std::string s = "dict";
        for (auto& kv : p->items) s += " " + (pmt::is_symbol(kv.first) ? kv.first->sym : show(kv.first)) + "=" + show(kv.second);
        return s;
    }
    default: return "nil";
    }
}

int main(int argc, char** argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage: %s cfg.txt iq.c64 outdir async seed max_noutput [propagate]\n", argv[0]); return 1; }
    // ---- configuration: "key v0 v1 ..." lines (integers / floats), "alist <path>" per code
    std::map<std::string, std::vector<std::string>> kv;
    std::vector<std::vector<int32_t>> occ_sets, pil_sets;
    std::vector<std::vector<float>> psym_rows;
    std::vector<std::string> alists;
    {
        std::ifstream f(argv[1]);
        std::string line;
        while (std::getline(f, line)) {
            auto w = split(line);
            if (w.empty()) continue;
            const std::string key = w[0];
            w.erase(w.begin());
            if (key == "occ") { std::vector<int32_t> s; for (auto& x : w) s.push_back(std::atoi(x.c_str())); occ_sets.push_back(s); }
            else if (key == "pilot_set") { std::vector<int32_t> s; for (auto& x : w) s.push_back(std::atoi(x.c_str())); pil_sets.push_back(s); }
            else if (key == "pilot_sym") { std::vector<float> s; for (auto& x : w) s.push_back(std::strtof(x.c_str(), nullptr)); psym_rows.push_back(s); }
            else if (key == "alist") { std::ifstream a(w[0]); std::stringstream ss; ss << a.rdbuf(); alists.push_back(ss.str()); }
            else kv[key] = w;
        }
    }
    auto geti = [&](const char* k) { return std::atoi(kv.at(k).at(0).c_str()); };
    auto getf = [&](const char* k) { return std::strtof(kv.at(k).at(0).c_str(), nullptr); };
    auto floats = [&](const char* k) { std::vector<float> v; for (auto& x : kv.at(k)) v.push_back(std::strtof(x.c_str(), nullptr)); return v; };
    const int N = geti("fft_len");
    std::vector<int32_t> occ_len, occ_flat, pil_flat;
    size_t max_occ = 0, n_pil = 1;
    for (auto& s : occ_sets) max_occ = std::max(max_occ, s.size());
    for (auto& s : occ_sets) { occ_len.push_back(int32_t(s.size())); for (size_t i = 0; i < max_occ; i++) occ_flat.push_back(i < s.size() ? s[i] : 0); }
    for (auto& s : pil_sets) n_pil = std::max(n_pil, s.size());
    for (auto& s : pil_sets) for (size_t i = 0; i < n_pil; i++) pil_flat.push_back(i < s.size() ? s[i] : int32_t(DTL_NO_CARRIER));
    std::vector<float> psym_flat;
    for (auto& r : psym_rows) for (size_t i = 0; i < 2 * n_pil; i++) psym_flat.push_back(i < r.size() ? r[i] : 0.0f);
    const std::vector<float> sw1 = floats("sync_word1"), sw2 = floats("sync_word2");
    if (int(sw1.size()) != 2 * N || int(sw2.size()) != 2 * N) { std::fprintf(stderr, "bad sync words\n"); return 1; }
    std::vector<const char*> texts;
    std::vector<size_t> lens;
    for (auto& a : alists) { texts.push_back(a.c_str()); lens.push_back(a.size()); }
    std::vector<double> mcs_snr;
    std::vector<int32_t> mcs_cnst, mcs_fec;
    for (auto& x : kv["mcs_snr"]) mcs_snr.push_back(std::strtod(x.c_str(), nullptr));
    for (auto& x : kv["mcs_cnst"]) mcs_cnst.push_back(std::atoi(x.c_str()));
    for (auto& x : kv["mcs_fec"]) mcs_fec.push_back(std::atoi(x.c_str()));
    dtl_rx_cfg cfg{};
    cfg.fft_len = N; cfg.cp_len = geti("cp_len"); cfg.frame_length = geti("frame_length"); cfg.has_fec = alists.empty() ? 0 : 1;
    cfg.n_occ_sets = int32_t(occ_sets.size()); cfg.max_occ = int32_t(max_occ); cfg.occ_len = occ_len.data(); cfg.occ = occ_flat.data();
    cfg.n_pilot_sets = int32_t(pil_sets.size()); cfg.n_pilots = int32_t(n_pil); cfg.pilot_carriers = pil_flat.data();
    cfg.n_pilot_sym_sets = int32_t(psym_rows.size()); cfg.pilot_symbols = psym_flat.data();
    cfg.sync_word1 = sw1.data(); cfg.sync_word2 = sw2.data();
    cfg.sync_threshold = getf("sync_threshold"); cfg.eq_alpha = 0.1f; cfg.use_sync_correct = geti("use_sync_correct");
    cfg.max_iterations = 15; cfg.n_fec_codes = int32_t(alists.size()); cfg.alist_texts = texts.data(); cfg.alist_lens = lens.data();
    cfg.n_mcs = int32_t(mcs_snr.size()); cfg.mcs_snr = mcs_snr.data(); cfg.mcs_cnst = mcs_cnst.data(); cfg.mcs_fec = mcs_fec.data();
    cfg.initial_mcs = geti("initial_mcs"); cfg.decision_th = 5; cfg.hysteresis = 1.0;
    cfg.device = DTL_DEVICE_CURRENT;

    // ---- the stream
    std::vector<gr_complex> x;
    {
        std::ifstream f(argv[2], std::ios::binary | std::ios::ate);
        const std::streamsize bytes = f.tellg();
        f.seekg(0);
        x.resize(size_t(bytes) / sizeof(gr_complex));
        f.read(reinterpret_cast<char*>(x.data()), bytes);
    }
    const std::string outdir = argv[3];
    const bool async = std::atoi(argv[4]) != 0;
    uint32_t lcg = uint32_t(std::atoi(argv[5])) * 2654435761u + 12345u;
    const int max_nout = std::atoi(argv[6]);
    const bool propagate = argc > 7 && std::atoi(argv[7]) != 0;
    auto rnd = [&](uint32_t m) { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 8) % m; };

    try {
        gr::dtl_hip::ofdm_adaptive_rx_hip::options opt;
        opt.len_tag_key = "packet_length";
        opt.async = async;
        opt.propagate_feedback_tags = propagate;
        opt.max_batch_samples = int64_t(1) << 20;
        auto blk = gr::dtl_hip::ofdm_adaptive_rx_hip::make(cfg, opt);
        gr::mock::harness h(blk);
        if (h.n_outputs() != 7 || h.out_itemsize(4) != size_t(N) * sizeof(gr_complex) || h.in_itemsize(0) != sizeof(gr_complex)) {
            std::fprintf(stderr, "io signature\n");
            return 2;
        }
        const int frame_samples = (2 + (cfg.has_fec ? 2 : 1) + cfg.frame_length) * (N + cfg.cp_len);
        size_t p = 0;
        long calls = 0, calls_short = 0;
        while (p < x.size()) {
            // ragged offers: a few samples, a fraction of a frame, several frames, now and then nothing at all
            int n_in;
            switch (rnd(6)) {
            case 0: n_in = int(rnd(64)); break;
            case 1: n_in = 257; break;
            case 2: n_in = frame_samples / 3 + int(rnd(50)); break;
            case 3: n_in = frame_samples + int(rnd(frame_samples)); break;
            default: n_in = 8 * frame_samples + int(rnd(1000)); break;
            }
            n_in = int(std::min<size_t>(size_t(n_in), x.size() - p));
            const int n_out = 1 + int(rnd(uint32_t(max_nout)));
            const int c = h.call(x.data() + p, n_in, n_out);
            if (c < n_in) calls_short++;
            p += size_t(c);
            calls++;
        }
        // everything accepted is decoded; then the scheduler keeps calling while the block asks for calls without input
        auto drain = [&]() {
            blk->wait_idle();
            for (int quiet = 0; quiet < 3;) {
                const uint64_t before = h.nitems_written(0) + h.nitems_written(2) + h.nitems_written(4) + h.nitems_written(5);
                (void)h.call(x.data(), 0, 1 + int(rnd(uint32_t(max_nout))));
                calls++;
                const uint64_t after = h.nitems_written(0) + h.nitems_written(2) + h.nitems_written(4) + h.nitems_written(5);
                quiet = after == before ? quiet + 1 : 0;
            }
        };
        drain();
        if (h.forecast(1000) != 1) { std::fprintf(stderr, "forecast: an idle block must ask for input\n"); return 3; }
        blk->stop();        // flush (final = 1): what it decodes is queued; a test may ask for it
        drain();
        // ---- write everything out
        for (size_t port = 0; port < 7; port++) {
            std::ofstream f(outdir + "/port" + std::to_string(port) + ".bin", std::ios::binary);
            f.write(reinterpret_cast<const char*>(h.captured(port).data()), std::streamsize(h.captured(port).size()));
        }
        {
            std::ofstream f(outdir + "/tags.txt");
            for (size_t port = 0; port < 7; port++)
                for (auto& t : h.tags(port)) f << port << " " << t.offset << " " << pmt::symbol_to_string(t.key) << " " << show(t.value) << "\n";
        }
        {
            std::ofstream f(outdir + "/msgs.txt");
            for (auto& m : h.messages()) f << pmt::symbol_to_string(m.first) << " " << show(m.second) << "\n";
        }
        std::printf("OK calls=%ld short=%ld read=%" PRIu64 " bytes=%" PRIu64 " vectors=%" PRIu64 " soft=%" PRIu64 " stream=%" PRIu64 " tags0=%zu msgs=%zu\n",
                    calls, calls_short, h.nitems_read(0), h.nitems_written(0), h.nitems_written(4), h.nitems_written(5),
                    h.nitems_written(2), h.tags(0).size(), h.messages().size());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 4;
    }
    return 0;
}
