// Self-test of tests/cpp/mock_gnuradio/ (no GPU, no libdtl_hip): a small block written against the stand-in headers is
// run by gr::mock::harness, and the harness must (a) account items, tags and messages the way a scheduler would and
// (b) catch a block that breaks the scheduler's rules (more produced than there was room for, more consumed than
// offered, a write behind the buffer, a tag behind nitems_written, a message on an unregistered port).
#include <gnuradio/block.h>

#include <cstdio>
#include <functional>
#include <string>

namespace {

// copies bytes, tags every 10th item with its index, publishes one message per call
class copy_block : public gr::block
{
public:
    int misbehave = 0;
    copy_block() : gr::block("copy", gr::io_signature::make(1, 1, 1), gr::io_signature::make(1, 1, 1)) { message_port_register_out(pmt::mp("info")); }
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items) override
    {
        const int n = std::min(noutput_items, ninput_items[0]);
        auto* in = static_cast<const unsigned char*>(input_items[0]);
        auto* out = static_cast<unsigned char*>(output_items[0]);
        for (int i = 0; i < n; i++) {
            out[i] = in[i];
            const uint64_t abs = nitems_written(0) + uint64_t(i);
            if (abs % 10 == 0) add_item_tag(0, abs, pmt::mp("idx"), pmt::from_long(long(abs)));
        }
        if (misbehave == 1) { consume_each(n); produce(0, noutput_items + 1); return WORK_CALLED_PRODUCE; }
        if (misbehave == 2) { consume_each(ninput_items[0] + 1); return n; }
        if (misbehave == 3) { out[noutput_items] = 1; }
        if (misbehave == 4) { add_item_tag(0, nitems_written(0) - 1, pmt::mp("idx"), pmt::from_long(0)); }
        if (misbehave == 5) { message_port_pub(pmt::mp("nope"), pmt::from_long(1)); }
        pmt::pmt_t m = pmt::dict_add(pmt::make_dict(), pmt::mp("n"), pmt::from_long(n));
        message_port_pub(pmt::mp("info"), m);
        consume_each(n);
        return n;
    }
};

bool throws(const std::function<void()>& f)
{
    try { f(); } catch (const std::exception&) { return true; }
    return false;
}

} // namespace

int main()
{
    std::vector<unsigned char> data(95);
    for (size_t i = 0; i < data.size(); i++) data[i] = (unsigned char)(i * 7);
    auto b = std::make_shared<copy_block>();
    gr::mock::harness h(b);
    size_t p = 0;
    int calls = 0;
    const int offers[] = { 0, 1, 13, 40, 7, 100 }, rooms[] = { 5, 3, 50, 8, 1, 60 };
    while (p < data.size()) {
        const int n_in = std::min<int>(offers[calls % 6], int(data.size() - p));
        p += size_t(h.call(data.data() + p, n_in, rooms[calls % 6]));
        calls++;
        if (calls > 1000) { std::printf("FAIL stuck\n"); return 1; }
    }
    if (h.captured(0) != data) { std::printf("FAIL data\n"); return 1; }
    if (h.nitems_read(0) != data.size() || h.nitems_written(0) != data.size()) { std::printf("FAIL counters\n"); return 1; }
    if (h.tags(0).size() != 10) { std::printf("FAIL tags %zu\n", h.tags(0).size()); return 1; }
    for (size_t i = 0; i < h.tags(0).size(); i++)
        if (h.tags(0)[i].offset != 10 * i || pmt::to_long(h.tags(0)[i].value) != long(10 * i) || h.tags(0)[i].key != pmt::mp("idx")) { std::printf("FAIL tag %zu\n", i); return 1; }
    if (int(h.messages().size()) != calls) { std::printf("FAIL messages\n"); return 1; }
    long total = 0;
    for (auto& m : h.messages()) total += pmt::to_long(pmt::dict_ref(m.second, pmt::mp("n"), pmt::from_long(-1000)));
    if (total != long(data.size())) { std::printf("FAIL message sum\n"); return 1; }
    // symbols are interned, dictionaries are values
    pmt::pmt_t d1 = pmt::dict_add(pmt::make_dict(), pmt::mp("a"), pmt::from_long(1));
    pmt::pmt_t d2 = pmt::dict_add(d1, pmt::mp("a"), pmt::from_long(2));
    if (pmt::mp("a") != pmt::string_to_symbol("a") || pmt::to_long(pmt::dict_ref(d1, pmt::mp("a"), PMT_NIL)) != 1 ||
        pmt::to_long(pmt::dict_ref(d2, pmt::mp("a"), PMT_NIL)) != 2 || pmt::length(d2) != 1) { std::printf("FAIL pmt\n"); return 1; }
    if (!throws([] { (void)pmt::to_long(pmt::from_double(1.0)); })) { std::printf("FAIL wrong_type\n"); return 1; }
    // rule breakers
    for (int kind = 1; kind <= 5; kind++) {
        auto bad = std::make_shared<copy_block>();
        gr::mock::harness hb(bad);
        (void)hb.call(data.data(), 20, 10);          // a clean first call: nitems_written > 0 afterwards
        bad->misbehave = kind;
        if (!throws([&] { (void)hb.call(data.data() + 10, 20, 10); })) { std::printf("FAIL misbehaviour %d not caught\n", kind); return 1; }
    }
    std::printf("OK calls=%d\n", calls);
    return 0;
}
