// rx_pipeline.cpp -- host orchestration of the batched receive path.
//
// The reference runs these steps as single-threaded GNU Radio blocks passing tags/messages; here
// the O(samples) arithmetic lives in HIP kernels (rx_kernels.hip, ldpc_bp.hip) and the O(frames)
// integer control logic -- which is inherently sequential and must be bit-exact -- runs on the
// host between kernel phases:
//   plateau_detector_fb            (GR, via ofdm_sync_sc_cfb)           -> plateau_from_edges
//   ofdm_adaptive_frame_detect_bb  (lib/dtl/..._frame_detect_bb_impl.cc) -> fix_sync_sparse
//   header_payload_demux + header_parser state (packet_header.cc:269-273) -> demux walk
//   tb_decoder::process_frame/decode (lib/dtl/tb_decoder.cc)            -> TbAssembler (gather tables)
//   ofdm_adaptive_feedback_decision + lost-frame counters               -> after phase B
#include "rx_pipeline.h"

#include "rx_kernels.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstring>
#include <condition_variable>
#include <initializer_list>
#include <limits>
#include <atomic>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <system_error>
#include <thread>

namespace dtl {

// ------------------------------------------------------------------ small helpers

DevBuf::~DevBuf()
{
    if (p) (void)hipFree(p);
}

int DevBuf::ensure(size_t bytes)
{
    if (bytes <= cap && p) return DTL_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = std::max<size_t>(bytes + bytes / 4, 256);
    DTL_HIP_CHECK(hipMalloc(&p, want));
    cap = want;
    return DTL_OK;
}

HostBuf::~HostBuf()
{
    if (p) (void)hipHostFree(p);
}

int HostBuf::ensure(size_t bytes)
{
    if (bytes <= cap && p && (mapped_coherent || !device_reads)) return DTL_OK;
    bytes = std::max(bytes, cap);                         // (a buffer that only changes its mapping keeps its size)
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    // device_reads: kernels read the buffer in place -- fine-grained (coherent) mapping, never cached on the device
    DTL_HIP_CHECK(hipHostMalloc(&p, want, device_reads ? (hipHostMallocMapped | hipHostMallocCoherent) : hipHostMallocDefault));
    mapped_coherent = device_reads;
    cap = want;
    return DTL_OK;
}

// the address kernels use for a buffer they read in place (HostBuf::device_reads)
static int device_view(const HostBuf& hb, void** out)
{
    DTL_HIP_CHECK(hipHostGetDevicePointer(out, hb.p, 0));
    return DTL_OK;
}

KernelClock::~KernelClock()
{
    for (auto& e : pool) if (e) (void)hipEventDestroy(e);
}

void KernelClock::begin(bool enable)
{
    on = enable;
    valid = false;
    if (!enable) return;
    used = 0;
    spans.clear();
}

int KernelClock::mark(hipStream_t st, size_t* idx)
{
    if (used == pool.size()) {
        hipEvent_t e = nullptr;
        DTL_HIP_CHECK(hipEventCreate(&e));
        pool.push_back(e);
    }
    DTL_HIP_CHECK(hipEventRecord(pool[used], st));
    *idx = used++;
    return DTL_OK;
}

int KernelClock::collect()
{
    for (int k = 0; k < N_KINDS; k++) { ms[k] = longest[k] = 0.0; launches[k] = 0; }
    for (const Span& sp : spans) {
        float t = 0.0f;
        DTL_HIP_CHECK(hipEventElapsedTime(&t, pool[sp.a], pool[sp.b]));
        ms[sp.kind] += double(t);
        longest[sp.kind] = std::max(longest[sp.kind], double(t));
        launches[sp.kind]++;
    }
    if (getenv("DTL_KC_TIMELINE") && !spans.empty()) {
        // (debugging aid) where the launches of the call sit on the device's clock, relative to the first one: the gaps
        // the profiler's own overhead would otherwise distort
        static const char* const names[N_KINDS] = { "sync_edges", "header", "frame_eq", "demap", "ldpc", "tb_check" };
        double busy_until = 0.0, idle = 0.0;
        for (const Span& sp : spans) {
            float a = 0.0f, b = 0.0f;
            DTL_HIP_CHECK(hipEventElapsedTime(&a, pool[spans[0].a], pool[sp.a]));
            DTL_HIP_CHECK(hipEventElapsedTime(&b, pool[spans[0].a], pool[sp.b]));
            const double gap = std::max(0.0, double(a) - busy_until);
            idle += gap;
            fprintf(stderr, "  kc %-10s start %8.3f dur %7.3f idle before %6.3f\n", names[sp.kind], double(a), double(b - a), gap);
            busy_until = std::max(busy_until, double(b));
        }
        fprintf(stderr, "  kc span %.3f ms, not covered by a timed launch %.3f ms\n", busy_until, idle);
    }
    valid = true;
    return DTL_OK;
}

// bracket one launch with the call's kernel clock (no-ops unless the call asked for the split)
#define KC_BEGIN(stream) size_t _kc_a = 0; if (kc_.on) DTL_TRY(kc_.mark((stream), &_kc_a))
#define KC_END(kind, stream) do { if (kc_.on) { size_t _kc_b = 0; DTL_TRY(kc_.mark((stream), &_kc_b)); kc_.span((kind), _kc_a, _kc_b); } } while (0)

int DeviceScope::enter(int device)
{
    if (device < 0) return DTL_OK;
    int cur = -1;
    DTL_HIP_CHECK(hipGetDevice(&cur));
    if (cur == device) return DTL_OK;
    DTL_HIP_CHECK(hipSetDevice(device));
    prev = cur;
    switched = true;
    return DTL_OK;
}

DeviceScope::~DeviceScope()
{
    if (switched) (void)hipSetDevice(prev);
}

int RxConfig::check_header_plan() const
{
    if (occupied_carriers.empty()) return DTL_OK;   // (reported by the caller's own checks)
    int sum = 0;
    for (int i = 0; i < header_len(); i++) sum += int(occupied_carriers[size_t(i) % occupied_carriers.size()].size());
    if (sum != header_bits()) {
        set_last_error("occupied_carriers: the carrier sets of the header symbols hold " + std::to_string(sum) + " bits, the header "
                       "formatter (header_len x len(occupied_carriers[0]), python/dtl/ofdm_receiver.py:129-130) " +
                       std::to_string(header_bits()) + ": the reference's header parser and demux would fall out of step");
        return DTL_ERR_INVALID_ARG;
    }
    return DTL_OK;
}

// H2D of a host vector through a pinned staging buffer
template <typename T>
static int upload_vec(HostBuf& hb, DevBuf& db, const std::vector<T>& v, hipStream_t st)
{
    const size_t bytes = v.size() * sizeof(T);
    if (!bytes) return DTL_OK;
    int rc = hb.ensure(bytes);
    if (rc != DTL_OK) return rc;
    rc = db.ensure(bytes);
    if (rc != DTL_OK) return rc;
    std::memcpy(hb.p, v.data(), bytes);
    DTL_HIP_CHECK(hipMemcpyAsync(db.p, hb.p, bytes, hipMemcpyHostToDevice, st));
    return DTL_OK;
}

#define DTL_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != DTL_OK) return _rc; \
    } while (0)

static int bits_per_symbol(int cnst) { return (cnst >= 1 && cnst <= 4) ? cnst : 0; }
static int compute_tb_len(int cw_len, int frame_len) { return frame_len > cw_len ? 1 + frame_len / cw_len : 1; }
static int align_bits_to_bytes(int nbits) { return nbits / 8 + ((nbits % 8) ? 1 : 0); }

static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------ plateau detector

std::vector<int64_t> plateau_from_edges(std::vector<unsigned long long> edges, int64_t n, int max_len)
{
    // the kernel appends edges with an atomic counter: sort them (LSD radix, 3 x 11 bits cover 2^33)
    if (edges.size() > 64) {
        std::vector<unsigned long long> tmp(edges.size());
        for (int pass = 0; pass < 3; pass++) {
            size_t cnt[2049] = { 0 };
            const int sh = pass * 11;
            for (unsigned long long e : edges) cnt[((e >> sh) & 2047) + 1]++;
            for (int b = 0; b < 2048; b++) cnt[b + 1] += cnt[b];
            for (unsigned long long e : edges) tmp[cnt[(e >> sh) & 2047]++] = e;
            edges.swap(tmp);
        }
    } else {
        std::sort(edges.begin(), edges.end());
    }
    // intervals [start, end) where metric >= threshold
    std::vector<std::pair<int64_t, int64_t>> runs;
    int64_t open = -1;
    for (unsigned long long e : edges) {
        const int64_t pos = int64_t(e >> 1);
        if (e & 1ull) { if (open < 0) open = pos; }
        else if (open >= 0) { runs.emplace_back(open, pos); open = -1; }
    }
    if (open >= 0) runs.emplace_back(open, n);
    std::vector<int64_t> trig;
    int64_t i = 0;
    size_t r = 0;
    while (i < n && r < runs.size()) {
        while (r < runs.size() && runs[r].second <= i) r++; // runs entirely behind the scan position
        if (r >= runs.size()) break;
        const int64_t flank_start = std::max(i, runs[r].first);
        i = runs[r].second; // first sample below threshold (or n)
        if (i - flank_start > 1) { // "1 sample is not a plateau"
            trig.push_back(flank_start + (i - flank_start) / 2);
            i = std::min(i + max_len, n - 1);
        }
        i++; // the for loop's increment
    }
    return trig;
}

// ------------------------------------------------------------------ frame_detect_bb::fix_sync

// The block's loop (one work() buffer of `len` items) as an event-driven walk over a sparse trigger list.  out[] of
// the reference is kept as two sorted lists: the input triggers (with erase marks) and the triggers the block inserts
// itself.  The scan position moves forward except after a correction that lands behind it, so cursors (re-seated by
// binary search only on a backward jump) replace a tree lookup per event.
//
// The walk is RESUMABLE: feed() takes the triggers as the detector produces them and runs every loop iteration whose
// outcome cannot depend on what is still unknown -- the triggers at or behind `horizon` and the buffer's length (which
// only enters through tests against positions near the buffer's end: the "missing trigger" deadline, "does the
// corrected position fit into this buffer", the remainder handed to the next buffer).  finish(len) runs the rest with
// the original conditions.  The receive path feeds one sample range at a time while the sync kernel works on the next
// (stage_sync), so that only the last range's share of the walk is left when the trigger list is complete.
FixSyncRun::FixSyncRun(FrameDetectState& st) : s(st)
{
    last_trigger_index = -int64_t(s.remainder);
    index = s.remainder;
}

void FixSyncRun::reseat(int64_t pos)
{
    cur = size_t(std::lower_bound(triggers.begin(), triggers.end(), pos) - triggers.begin());
    cur_add = size_t(std::lower_bound(added.begin(), added.end(), pos) - added.begin());
}

int64_t FixSyncRun::next_set(int64_t pos)
{
    const size_t nt = triggers.size();
    while (cur < nt && (triggers[cur] < pos || erased[cur])) cur++;
    while (cur_add < added.size() && added[cur_add] < pos) cur_add++;
    int64_t t = kInf;
    if (cur < nt) t = triggers[cur];
    if (cur_add < added.size()) t = std::min(t, added[cur_add]);
    return t;
}

void FixSyncRun::insert_added(int64_t pos)
{
    auto it = std::lower_bound(added.begin(), added.end(), pos);
    if (it == added.end() || *it != pos) {
        const size_t at = size_t(it - added.begin());
        added.insert(it, pos);
        if (at < cur_add) cur_add++; // keep the cursor on the same element
    }
}

void FixSyncRun::erase_at(int64_t pos)
{
    auto it = std::lower_bound(triggers.begin(), triggers.end(), pos);
    if (it != triggers.end() && *it == pos) erased[size_t(it - triggers.begin())] = 1;
    auto ia = std::lower_bound(added.begin(), added.end(), pos);
    if (ia != added.end() && *ia == pos) {
        const size_t at = size_t(ia - added.begin());
        added.erase(ia);
        if (at < cur_add) cur_add--;
    }
}

void FixSyncRun::feed(const int64_t* trig, size_t n, int64_t horizon_)
{
    triggers.insert(triggers.end(), trig, trig + n);
    erased.resize(triggers.size(), 0);
    if (horizon_ > horizon) horizon = horizon_;
    advance(false, 0);
}

// FINAL: every condition as the reference writes it, `len` known.  Otherwise an iteration runs only when its outcome
// is the same for every len >= horizon and whatever triggers lie at or behind the horizon; the first one that is not
// stops the walk (nothing of it has been applied yet).
void FixSyncRun::advance(bool final, int64_t len)
{
    constexpr int kSyncedTh = 3, kMaxMissing = 5; // :21-22
    const int64_t FL = s.frame_len, H = horizon;
    while (final ? i < len : i < H) {
        int64_t t = next_set(i);
        if (t >= (final ? len : H)) t = kInf;
        int64_t miss = kInf;
        bool miss_known = true;
        if (s.in_sync) {
            const int64_t m0 = i + std::max<int64_t>(0, (FL + 11) - index);
            if (final) {
                miss = m0;
                if (index + (len - 1 - i) > FL + 1) miss = std::min(miss, len - 1);
                if (miss >= len) miss = kInf;
            } else if (m0 < H) {
                miss = m0;                 // m0 <= len - 1: neither the clamp to len - 1 nor the cut at len acts
            } else {
                miss_known = false;        // some position >= H - 1, or none: not before every known trigger's turn
            }
        }
        if (!final && t == kInf && (!miss_known || !s.in_sync)) break;   // the next event lies in what is not known yet
        // (a known trigger t < H with an unknown deadline: the deadline is >= H - 1 >= t, and a tie is the trigger's)
        const int64_t ev = std::min(t, miss_known ? miss : kInf);
        if (ev == kInf) break;
        if (t == ev) {
            const int64_t diff = (ev - last_trigger_index) - FL;
            const int64_t inst_error = diff < 0 ? -diff : diff;
            const int acc_after = s.acc_error + (inst_error == 1 ? int(diff) : 0);
            const bool corrects = acc_after > 1 || (inst_error > 1 && inst_error < 10);
            if (!final && corrects && !(last_trigger_index + FL < H)) break; // "fits into this buffer" needs len
            index += ev - i;
            i = ev;
            index = 0;
            if (s.acc_error) ++s.trigger_counter;
            s.acc_error = acc_after;
            s.error_count += inst_error != 0;
            if (inst_error == 0 || inst_error == 1) {
                if (s.synced < kSyncedTh) ++s.synced;
                else s.in_sync = true;
            } else {
                s.synced = 0;
                s.in_sync = false;
            }
            if (corrects) {
                if (last_trigger_index + FL < (final ? len : H)) {
                    erase_at(i);
                    last_trigger_index += FL;
                    if (last_trigger_index >= 0) insert_added(last_trigger_index);
                    const bool backward = last_trigger_index < i;
                    i = last_trigger_index;
                    if (backward) reseat(i + 1);
                    s.acc_error = 0;
                    ++s.correction_count;
                } else {
                    last_trigger_index = i;
                }
            } else {
                last_trigger_index = i;
            }
            trigger_found = true;
            if (s.trigger_counter > 10) s.acc_error = 0;
        } else {
            index += ev - i;
            i = ev;
            last_trigger_index += FL;
            if (last_trigger_index >= 0) insert_added(last_trigger_index);
            const bool backward = last_trigger_index < i;
            i = last_trigger_index;
            if (backward) reseat(i + 1);
            index = 0;
            trigger_found = true;
            ++s.missing_count;
            if (--s.synced < kSyncedTh - kMaxMissing) {
                s.in_sync = false;
                s.synced = 0;
            }
        }
        ++i;
        ++index;
    }
}

std::vector<int64_t> FixSyncRun::finish(int64_t len)
{
    advance(true, len);
    // d_remainder: the distance from the last trigger to the end of the buffer (assigned at every trigger event in the
    // reference; only the last assignment survives)
    if (trigger_found) s.remainder = int(len - last_trigger_index);
    else s.remainder += int(len);
    // merge the two lists
    const size_t nt = triggers.size();
    std::vector<int64_t> res;
    res.reserve(nt + added.size());
    size_t a = 0, b2 = 0;
    while (a < nt || b2 < added.size()) {
        while (a < nt && erased[a]) a++;
        const int64_t va = a < nt ? triggers[a] : kInf, vb = b2 < added.size() ? added[b2] : kInf;
        if (va == kInf && vb == kInf) break;
        if (va < vb) { if (va >= 0 && va < len) res.push_back(va); a++; }
        else if (vb < va) { if (vb >= 0 && vb < len) res.push_back(vb); b2++; }
        else { if (va >= 0 && va < len) res.push_back(va); a++; b2++; }
    }
    return res;
}

std::vector<int64_t> fix_sync_sparse(FrameDetectState& s, const std::vector<int64_t>& triggers, int64_t len)
{
    FixSyncRun run(s);
    run.feed(triggers.data(), triggers.size(), 0);   // horizon 0: nothing is decided before the length is known
    return run.finish(len);
}

// ------------------------------------------------------------------ feedback decision

void FeedbackDecision::init(const std::vector<McsEntry>& mcs, double hyst, int th, int initial)
{
    lut = mcs;
    if (lut.empty()) lut.push_back({ 0.0, 1, 0 });
    lut[0].snr_th = DBL_MIN; // feedback_decision.cc:50
    hysteresis = hyst;
    decision_th = th;
    counter = 0;
    last = pending = std::min<int>(std::max(initial, 0), int(lut.size()) - 1);
}

void FeedbackDecision::get(double snr, int* cnst, int* fec)
{
    auto update = [&](int id) {
        if (id != pending) { counter = 0; pending = id; }
        else if (++counter >= decision_th) { counter = 0; last = pending; }
    };
    const int cur = last;
    if (snr < lut[cur].snr_th) update(cur > 0 ? cur - 1 : 0); // reference asserts cur > 0 (:66)
    else if (cur + 1 < int(lut.size())) {
        if (snr > lut[cur + 1].snr_th + hysteresis) update(cur + 1);
        else counter = 0;
    } else counter = 0;
    *cnst = lut[last].cnst;
    *fec = lut[last].fec;
}

// ------------------------------------------------------------------ transport-block reassembly

namespace {

// tb_decoder (lib/dtl/tb_decoder.cc) with the receive buffer expressed as source segments of the
// soft-bit stream instead of copied floats; decode() emits one gather recipe per TB, the per-codeword
// descriptors are expanded on the device (expand_cw_kernel).
struct TbAssembler {
    struct Seg { uint32_t off; int len; };
    struct Done { TbGather g; int code, tb_no, tb_payload; std::vector<Seg> pieces; }; // pieces: only of an irregular block
    std::vector<Seg> rcv;
    long rcv_size = 0, rcv_cap = 0, buf_idx = 0;
    int tb_number = -1;
    bool have_info = false;
    int info_tb_payload = 0, info_code = 0;
    std::vector<Done> done;
    bool irregular = false; // a block needs more pieces than a TbGather / CwDesc describes (Done::pieces holds them)

    void append(uint32_t base, long lo, long hi, long in_len)
    {
        if (hi <= lo) return;
        // reads past the frame's soft bits (reference: whatever follows in the buffer) are
        // defined as "never received": a gap segment, decoded as soft value 0
        const long h2 = std::min(hi, in_len);
        if (h2 > lo) {
            // a piece that continues the previous one in the soft-bit stream (consecutive frames used to their last
            // bit) extends it: a transport block over many short frames stays one segment
            if (!rcv.empty() && rcv.back().off != 0xFFFFFFFFu && uint64_t(rcv.back().off) + uint64_t(rcv.back().len) == uint64_t(base + lo))
                rcv.back().len += int(h2 - lo);
            else
                rcv.push_back({ uint32_t(base + lo), int(h2 - lo) });
        }
        if (hi > std::max(h2, lo)) rcv.push_back({ 0xFFFFFFFFu, int(hi - std::max(h2, lo)) });
        rcv_size += hi - lo;
    }
    void clear() { rcv.clear(); rcv_size = 0; }

    void decode(int tb_len, int n, int k, int tb_no, int tb_payload)
    {
        Done d{};
        d.code = info_code; d.tb_no = tb_no; d.tb_payload = tb_payload;
        TbGather& g = d.g;
        g.n_seg = int16_t(std::min<size_t>(rcv.size(), 4));
        // A gather recipe holds 4 pieces and a codeword's descriptor 2: enough when frames are used to their last bit
        // (consecutive frames are one piece) and a loss or a constellation change splits a block now and then.  Frames
        // whose bit capacity is not a whole number of bytes leave a gap after every frame: such a block keeps its piece
        // list and is copied into one contiguous run first (stage_blocks) -- the reference's RCV_BUF, only for these.
        bool many = rcv.size() > 4;
        for (size_t i = 1; i + 1 < rcv.size(); i++) many = many || rcv[i].len < n; // a codeword could span 3 pieces
        if (many) { d.pieces = rcv; irregular = true; }
        for (int i = 0; i < 4; i++) {
            g.seg_off[i] = i < g.n_seg ? rcv[i].off : 0xFFFFFFFFu;
            g.seg_len[i] = i < g.n_seg ? rcv[i].len : 0;
        }
        g.tb_payload = std::max(info_tb_payload, 0);
        g.n_cw = int16_t(tb_len);
        g.ncheck = int16_t(n - k);
        g.k = int16_t(k);
        done.push_back(d);
    }
};

struct CodeDims { int n = 0, k = 0; };

// One frame through ofdm_adaptive_fec_decoder_impl::general_work (:80-217) -> tb_decoder::process_frame
// (tb_decoder.cc:32-141), on index ranges of the soft-bit stream: `base` = stream offset of the frame's first soft
// bit, n_syms x bps of them present.  dims[fec id] = (n, k) of the code; completed blocks are appended to asmb.done.
int tb_process_frame(TbAssembler& asmb, const FrameInfo& fi, int cnst, int n_syms, uint32_t base, int capacity,
                     const std::vector<CodeDims>& dims)
{
    if (fi.payload_len == 0) return DTL_OK; // empty frame never reaches the FEC decoder
    const int bps = bits_per_symbol(cnst);
    const long in_len = long(n_syms) * bps;
    if (fi.fec <= 0 || fi.fec >= int(dims.size()) || dims[fi.fec].n <= 0) return DTL_OK; // reference dereferences a null decoder here
    const int n_ = dims[fi.fec].n, k_ = dims[fi.fec].k, ncheck = n_ - k_;
    const long frame_payload = 8L * fi.payload_len;           // fec_utils.cc:93
    const long tb_offset = 8L * fi.tb_offset;                 // fec_decoder_impl.cc:147
    const int frame_len = 8 * align_bits_to_bytes(capacity * bps); // :207
    auto info_nk = [&](int& nn, int& kk) { nn = dims[asmb.info_code].n; kk = dims[asmb.info_code].k; };
    if (fi.tb_no == asmb.tb_number) {
        if (asmb.buf_idx + frame_payload > asmb.rcv_cap) { set_last_error("rcv buffer does not have enough capacity!"); return DTL_ERR_CAPACITY; }
        asmb.append(base, 0, frame_payload, in_len);
        asmb.buf_idx += frame_payload;
        const int ncws = compute_tb_len(n_, frame_len);
        if (asmb.rcv_size >= long(ncheck) * ncws + fi.tb_payload) {
            int nn, kk; info_nk(nn, kk);
            asmb.decode(ncws, nn, kk, fi.tb_no, fi.tb_payload);
            asmb.buf_idx = 0; asmb.clear();
        }
    } else if (asmb.rcv_size == 0 && tb_offset == frame_payload) {
        asmb.have_info = true; asmb.info_tb_payload = fi.tb_payload; asmb.info_code = fi.fec;
        asmb.tb_number = fi.tb_no; asmb.buf_idx = 0; asmb.clear();
        asmb.append(base, 0, tb_offset, in_len);
        asmb.decode(compute_tb_len(n_, frame_len), n_, k_, fi.tb_no, fi.tb_payload);
        asmb.clear();
    } else {
        if (tb_offset > 0) asmb.append(base, 0, tb_offset, in_len);
        if (asmb.rcv_size > 0 && asmb.have_info) {
            int nn, kk; info_nk(nn, kk);
            asmb.decode(compute_tb_len(nn, frame_len), nn, kk, asmb.tb_number, asmb.info_tb_payload);
        }
        asmb.have_info = true; asmb.info_tb_payload = fi.tb_payload; asmb.info_code = fi.fec;
        asmb.tb_number = fi.tb_no; asmb.buf_idx = 0; asmb.clear();
        int extra_bits = 0;
        if (tb_offset % bps) extra_bits = int(bps - tb_offset % bps);
        const long new_tb_offset = tb_offset ? tb_offset + extra_bits : 0;
        asmb.append(base, new_tb_offset, frame_payload + extra_bits, in_len);
        asmb.buf_idx += frame_payload - tb_offset;
        const int tb_len = compute_tb_len(n_, frame_len);
        const long tb_size = 8L * align_bits_to_bytes(tb_len * ncheck + fi.tb_payload);
        if (frame_payload - tb_offset == tb_size) {
            asmb.decode(tb_len, n_, k_, fi.tb_no, fi.tb_payload);
            asmb.buf_idx = 0; asmb.clear();
        }
    }
    return DTL_OK;
}

} // namespace

// dtl_tb_gather (C ABI): the transport-block reassembly alone, for a self-contained run of frames.  -> gather recipes
// (one per completed block, with cw_first / bits_off filled in stream order) and the code each block uses.
int tb_gather_frames(const std::vector<std::pair<int, int>>& code_nk, int capacity, int max_bps, const FrameInfo* frames,
                     int64_t n_frames, std::vector<TbGather>& gathers, std::vector<TbSpan>& spans)
{
    TbAssembler asmb;
    std::vector<CodeDims> dims(code_nk.size() + 1);
    int max_n = 0;
    for (size_t i = 0; i < code_nk.size(); i++) { dims[i + 1] = { code_nk[i].first, code_nk[i].second }; max_n = std::max(max_n, code_nk[i].first); }
    asmb.rcv_cap = 2L * max_n * compute_tb_len(max_n, capacity * max_bps); // tb_decoder.cc:24-27
    for (int64_t f = 0; f < n_frames; f++) {
        const FrameInfo& fi = frames[f];
        if (!fi.header_ok || fi.llr_offset < 0) continue;
        const int bps = bits_per_symbol(fi.cnst);
        if (bps <= 0) { set_last_error("Tags missing"); return DTL_ERR_TAGS_MISSING; } // fec_decoder_impl.cc:121-126
        const int n_syms = fi.payload_len ? std::min(fi.n_payload_syms, capacity) : 0;
        DTL_TRY(tb_process_frame(asmb, fi, fi.cnst, n_syms, uint32_t(fi.llr_offset), capacity, dims));
    }
    if (asmb.irregular) { set_last_error("transport block spread over more pieces than a gather recipe holds (the whole-chain call copies such blocks; dtl_tb_gather does not own the soft-bit stream)"); return DTL_ERR_UNSUPPORTED; }
    gathers.clear(); spans.clear();
    uint32_t bits_base = 0, cw_first = 0;
    for (auto& d : asmb.done) {
        d.g.bits_off = bits_base;
        d.g.cw_first = cw_first;
        bits_base += (uint32_t(d.g.tb_payload) + 15u) & ~7u;
        gathers.push_back(d.g);
        spans.push_back({ d.tb_no, d.code, d.g.tb_payload, int(d.g.n_cw), cw_first, d.g.bits_off });
        cw_first += uint32_t(d.g.n_cw);
    }
    return DTL_OK;
}

// ------------------------------------------------------------------ stream state

// Everything a continued stream needs from the calls before it.  Positions are absolute sample
// indices (0 = first sample after reset_stream()).
struct Receiver::StreamState {
    int64_t next_abs = 0;                        // absolute index of the next new sample
    int64_t carry_abs = 0, carry_len = 0;        // d_carry_ holds samples [carry_abs, carry_abs + carry_len)
    std::vector<unsigned long long> edge_carry;  // rising edge of a plateau that is still open
    int64_t plateau_i = 0;                       // plateau detector scan position
    int64_t trig_produced = 0;                   // frame_detect_bb has seen the trigger stream up to here
    std::vector<CfoSegment> segs;                // oscillator segments still needed (absolute starts)
    FrameDetectState fds;
    std::vector<int64_t> pending_fixed;          // triggers whose frame is not complete yet
    int64_t demux_pos = 0;                       // header_payload_demux: samples consumed
    int state_cnst = 4;                          // header parser: d_constellation(QAM16), packet_header.cc:70
    TbAssembler asmb;                            // transport block being received
    int64_t llr_carry_len = 0;                   // its soft bits, compacted in d_llr_carry_
    FeedbackDecision fb;
    bool fb_init = false;
    int expected_frame_no = 0;
    long lost_total = 0;
};

// The sync stage's helper thread (stage_sync): ONE thread per receiver, started when the first call needs it and kept for
// the handle's lifetime (a std::thread per call cost its creation and join inside every step, and its constructor can
// throw std::system_error).  run() hands over a job, wait() returns when the job has finished; a job never lets an
// exception out (it reports through its own result fields).
class Receiver::Worker {
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker()
    {
        { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    void run(std::function<void()> job)
    {
        { std::lock_guard<std::mutex> lk(mu_); job_ = std::move(job); has_job_ = true; done_ = false; }
        cv_.notify_all();
    }
    void wait() noexcept
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return done_; });
    }

private:
    void loop()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return quit_ || has_job_; });
                if (quit_) return;
                job = std::move(job_);
                has_job_ = false;
            }
            try { job(); } catch (...) { /* jobs report through their own result fields */ }
            { std::lock_guard<std::mutex> lk(mu_); done_ = true; }
            cv_done_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    std::function<void()> job_;
    bool has_job_ = false, done_ = true, quit_ = false;
    std::thread th_;    // last member: the thread starts when everything it touches exists
};

Receiver::Receiver() : ss_(std::make_unique<StreamState>())
{
    // tests lower the threshold so that small streams take the pipelined (four ranges, early header launches) path
    if (const char* e = getenv("DTL_RX_SPLIT_MIN")) split_min_ = std::max<int64_t>(1, atoll(e));
    if (const char* e = getenv("DTL_RX_SYNC_RANGES")) sync_ranges_ = std::min(16, std::max(1, atoi(e)));
    if (const char* e = getenv("DTL_RX_ZERO_COPY_DESCS")) zero_copy_descs_ = atoi(e) != 0;
    if (const char* e = getenv("DTL_RX_ZERO_COPY_CANDS")) zero_copy_cands_ = atoi(e) != 0;
    if (const char* e = getenv("DTL_RX_ZERO_COPY_TRIG")) zero_copy_trig_ = atoi(e) != 0;
    if (const char* e = getenv("DTL_RX_FLUSH_FRAMES")) flush_frames_ = std::max(64, atoi(e));   // (tuning experiments)
    if (const char* e = getenv("DTL_RX_FLUSH_FIRST")) flush_first_ = std::max(64, atoi(e));
}

void Receiver::reset_stream() noexcept
{
    if (side_) (void)hipStreamSynchronize(side_); // a failed call may leave trigger-angle work in flight
    // in place (no allocation: this runs from destructors while an exception unwinds the failed call)
    *ss_ = StreamState{};
}

int64_t Receiver::stream_position() const { return ss_->next_abs; }

int Receiver::kernel_times(int back, double* ldpc_ms, double* payload_ms)
{
    if (ldpc_ms) *ldpc_ms = -1.0;
    if (payload_ms) *payload_ms = -1.0;
    if (back < 0 || back >= kEvRing || uint64_t(back) >= n_calls_) { set_last_error("kernel_times: no such call in the ring"); return DTL_ERR_INVALID_ARG; }
    const int slot = int((n_calls_ - 1 - uint64_t(back)) % kEvRing);
    hipEvent_t* ev = ev_ring_[slot];
    float ms = 0;
    if ((ev_mask_[slot] & 1u) && payload_ms && hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) *payload_ms = ms;
    if ((ev_mask_[slot] & 2u) && ldpc_ms && hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) *ldpc_ms = ms;
    return DTL_OK;
}

// ------------------------------------------------------------------ Receiver

Receiver::~Receiver()
{
    for (auto& slot : ev_ring_) for (auto& e : slot) if (e) (void)hipEventDestroy(e);
    for (auto& e : chunk_ev_) if (e) (void)hipEventDestroy(e);
    if (ev_stats_) (void)hipEventDestroy(ev_stats_);
    for (auto& e : angle_ev_) if (e) (void)hipEventDestroy(e);
    for (auto& e : hdr_ev_) if (e) (void)hipEventDestroy(e);
    if (side_) { (void)hipStreamSynchronize(side_); (void)hipStreamDestroy(side_); }
    if (ev_pre_) (void)hipEventDestroy(ev_pre_);
    if (decoder_gate) { std::lock_guard<std::mutex> lk(decoder_gate->mu); for (hipEvent_t e : ev_dec_) if (decoder_gate->last_done == e) decoder_gate->last_done = nullptr; }
    for (hipEvent_t e : ev_dec_) if (e) (void)hipEventDestroy(e);
}

int Receiver::init()
{
    const int N = cfg.fft_len;
    if (N < 64 || N > 1024 || (N & (N - 1))) { set_last_error("fft_len must be a power of two in [64,1024]"); return DTL_ERR_INVALID_ARG; }
    if (cfg.cp_len < 0 || cfg.cp_len > N) { set_last_error("bad cp_len"); return DTL_ERR_INVALID_ARG; }
    if (cfg.frame_length < 1) { set_last_error("bad frame_length"); return DTL_ERR_INVALID_ARG; }
    if (cfg.occupied_carriers.empty() || cfg.occupied_carriers[0].empty()) { set_last_error("occupied_carriers empty"); return DTL_ERR_INVALID_ARG; }
    if (int(cfg.sync_word1.size()) != N || int(cfg.sync_word2.size()) != N) {
        set_last_error("Length of sync sequence(s) must be FFT length."); // ofdm_receiver.py:52-54
        return DTL_ERR_INVALID_ARG;
    }
    for (auto& set : cfg.occupied_carriers) {
        std::vector<uint8_t> seen(size_t(N), 0);
        for (int c : set) {
            int idx = c < 0 ? c + N : c;
            if (idx < 0 || idx >= N) { set_last_error("data carrier index out of bounds"); return DTL_ERR_INVALID_ARG; }
            if (seen[idx]) { set_last_error("data carrier listed twice in one set"); return DTL_ERR_INVALID_ARG; }
            seen[idx] = 1;
        }
    }
    const size_t np = cfg.pilot_carriers.empty() ? 0 : cfg.pilot_carriers[0].size();
    for (auto& set : cfg.pilot_carriers)
        if (set.size() != np) { set_last_error("pilot carrier sets must have equal size"); return DTL_ERR_INVALID_ARG; }
    for (auto& set : cfg.pilot_symbols)
        if (set.size() != np) { set_last_error("pilot symbol sets must match the pilot carriers"); return DTL_ERR_INVALID_ARG; }
    if (!cfg.pilot_carriers.empty() && cfg.pilot_symbols.empty()) { set_last_error("pilot_symbols empty"); return DTL_ERR_INVALID_ARG; }
    DTL_TRY(cfg.check_header_plan());
    capacity_ = 0;
    for (int i = 0; i < cfg.frame_length; i++) capacity_ += int(cfg.occupied_carriers[i % cfg.occupied_carriers.size()].size());
    decoders_.clear();
    decoders_.push_back(nullptr); // index 0: "no FEC" (ldpc_dec.cc:25)
    for (auto& text : cfg.alist_texts) {
        auto dec = std::make_unique<LdpcDecoder>();
        std::string err = dec->code.parse_alist(text);
        if (!err.empty()) { set_last_error(err); return DTL_ERR_INVALID_ARG; }
        dec->code.reduce();
        dec->code.build_schedules();
        dec->max_iterations = cfg.max_iterations > 0 ? cfg.max_iterations : 15;
        if (cfg.bp_sigma > 0) dec->sigma = cfg.bp_sigma;
        dec->arith = cfg.arith;
        decoders_.push_back(std::move(dec));
    }
    // (a handle with has_fec but without codes serves the single-block entry points -- header length 2 -- and is
    // refused by process(): ofdm_adaptive_fec_decoder_impl.cc "Misconfiguration: decoder not found!")
    if (cfg.mcs.empty()) cfg.mcs = { { DBL_MIN, 1, 0 }, { 13, 2, 0 }, { 18, 3, 0 }, { 23, 4, 0 } }; // config.py:43-44
    return DTL_OK;
}

int Receiver::upload_tables()
{
    const int N = cfg.fft_len;
    const int h = cfg.header_len();
    auto shifted = [&](int c) { int idx = c < 0 ? c + N : c; return (idx + N / 2) % N; };
    std::vector<uint8_t> blob;
    auto put = [&](const void* p, size_t bytes) {
        size_t off = (blob.size() + 15) & ~size_t(15);
        blob.resize(off + bytes);
        if (bytes) std::memcpy(blob.data() + off, p, bytes);
        return off;
    };
    std::vector<cf32> tw_nat(N / 2);
    for (int k = 0; k < N / 2; k++) {
        const double ang = -2.0 * M_PI * k / N;
        tw_nat[k] = { float(std::cos(ang)), float(std::sin(ang)) };
    }
    int logn = 0;
    while ((1 << logn) < N) logn++;
    // The FFT of the receive kernels runs two butterfly stages per pass (load_and_fft4); pass `st` (h = 2^st) reads
    // W^(k N/2h), W^(k N/4h), W^((k+h) N/4h) for k < h.  Stored contiguously per pass -- [2^st - 1 + k],
    // [.. + h + k], [.. + 2h + k] -- consecutive lanes read consecutive entries (the natural table would be read
    // with strides N/2h, N/4h: the same LDS banks).  An odd last stage reads the natural table, appended at 2^st - 1.
    std::vector<cf32> tw(size_t(N), cf32{ 0.0f, 0.0f });
    {
        int st = 0;
        for (; st + 1 < logn; st += 2) {
            const int hh = 1 << st, base = hh - 1;
            for (int k = 0; k < hh; k++) {
                tw[base + k] = tw_nat[k << (logn - st - 1)];
                tw[base + hh + k] = tw_nat[k << (logn - st - 2)];
                tw[base + 2 * hh + k] = tw_nat[(k + hh) << (logn - st - 2)];
            }
        }
        if (st < logn)
            for (int k = 0; k < N / 2; k++) tw[(1 << st) - 1 + k] = tw_nat[k];
    }
    std::vector<uint16_t> br(N);
    for (int i = 0; i < N; i++) {
        int r = 0;
        for (int b = 0; b < logn; b++) if (i & (1 << b)) r |= 1 << (logn - 1 - b);
        br[i] = uint16_t(fft_lds_swizzle(r));
    }
    const int n_occ_sets = int(cfg.occupied_carriers.size());
    int max_occ = 0;
    for (auto& s : cfg.occupied_carriers) max_occ = std::max<int>(max_occ, int(s.size()));
    std::vector<int16_t> occ_idx(size_t(n_occ_sets) * max_occ, 0), occ_len(n_occ_sets);
    std::vector<uint8_t> occupied(N, 0);
    std::vector<int16_t> occ_rank(size_t(n_occ_sets) * N, -1);
    for (int s = 0; s < n_occ_sets; s++) {
        occ_len[s] = int16_t(cfg.occupied_carriers[s].size());
        for (size_t k = 0; k < cfg.occupied_carriers[s].size(); k++) {
            const int idx = shifted(cfg.occupied_carriers[s][k]);
            occ_idx[size_t(s) * max_occ + k] = int16_t(idx);
            occupied[idx] = 1;
        }
        // a carrier listed twice would be read by both serializer slots; the equaliser lane can write one of them
        // (first listing wins here): such maps are rejected in init()
        for (size_t k = cfg.occupied_carriers[s].size(); k-- > 0;)
            occ_rank[size_t(s) * N + shifted(cfg.occupied_carriers[s][k])] = int16_t(k);
    }
    const int n_pilot_sets = int(cfg.pilot_carriers.size());
    const int n_pilots = n_pilot_sets ? int(cfg.pilot_carriers[0].size()) : 0;
    const int n_psym = std::max<int>(1, int(cfg.pilot_symbols.size()));
    std::vector<uint8_t> pmask(size_t(std::max(n_pilot_sets, 1)) * N, 0);
    std::vector<cf32> pval(size_t(n_psym) * N, cf32{ 0, 0 });
    for (int s = 0; s < n_pilot_sets; s++)
        for (int k = 0; k < n_pilots; k++)
            if (cfg.pilot_carriers[s][k] != DTL_NO_CARRIER) pmask[size_t(s) * N + shifted(cfg.pilot_carriers[s][k])] = 1;
    for (size_t s = 0; s < cfg.pilot_symbols.size(); s++) { // equalizer.cc:196-214: all symbol sets are loaded
        const int cset = int(s % std::max(n_pilot_sets, 1));
        for (int k = 0; k < n_pilots; k++)
            if (cfg.pilot_carriers[cset][k] != DTL_NO_CARRIER) pval[s * N + shifted(cfg.pilot_carriers[cset][k])] = cfg.pilot_symbols[s][k];
    }
    // ofdm_chanest_vcvc ctor
    std::vector<cf32> corr(N), ref(cfg.sync_word2);
    int first_active = 0, last_active = N - 1;
    for (int i = 0; i < N; i++) if (ref[i].re != 0 || ref[i].im != 0) { first_active = i; break; }
    for (int i = N - 1; i >= 0; i--) if (ref[i].re != 0 || ref[i].im != 0) { last_active = i; break; }
    int gmin = -first_active, gmax = N - last_active - 1;
    if (gmin % 2) gmin++;
    if (gmax % 2) gmax--;
    for (int i = 0; i < N; i++) {
        const cf32 s1 = cfg.sync_word1[i], s2 = cfg.sync_word2[i];
        if (s1.re == 0 && s1.im == 0) corr[i] = { 0, 0 };
        else { // complex division, Smith's form (what std::complex<float> /= does through libgcc)
            const float a = s2.re, b = s2.im, c = s1.re, d = s1.im;
            if (std::fabs(c) < std::fabs(d)) { float r = c / d, den = (c * r) + d; corr[i] = { ((a * r) + b) / den, ((b * r) - a) / den }; }
            else { float r = d / c, den = (d * r) + c; corr[i] = { ((b * r) + a) / den, (b - (a * r)) / den }; }
        }
    }
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
    const size_t o_oi = put(occ_idx.data(), occ_idx.size() * 2);
    const size_t o_ork = put(occ_rank.data(), occ_rank.size() * 2);
    const size_t o_ol = put(occ_len.data(), occ_len.size() * 2);
    const size_t o_oc = put(occupied.data(), occupied.size());
    const size_t o_pm = put(pmask.data(), pmask.size());
    const size_t o_pv = put(pval.data(), pval.size() * sizeof(cf32));
    const size_t o_cv = put(corr.data(), corr.size() * sizeof(cf32));
    std::vector<uint16_t> sync_act; // carriers that take part in the integer-offset search (corr_v != 0), ascending
    for (int i = 0; i < N; i++) if (corr[i].re != 0 || corr[i].im != 0) sync_act.push_back(uint16_t(i));
    const size_t o_sa = put(sync_act.data(), std::max<size_t>(sync_act.size(), 1) * 2);
    const size_t o_rf = put(ref.data(), ref.size() * sizeof(cf32));
    const size_t o_pt = put(pts.data(), pts.size() * sizeof(cf32));
    std::vector<double> recip(size_t(cfg.frame_length) * N + 1, 0.0); // every pilot sample of a frame counted once
    for (size_t k = 1; k < recip.size(); k++) recip[k] = 1.0 / double(k);
    const size_t o_rc = put(recip.data(), recip.size() * sizeof(double));
    int rc = tab_blob_.ensure(blob.size());
    if (rc != DTL_OK) return rc;
    DTL_HIP_CHECK(hipMemcpy(tab_blob_.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    auto* b = static_cast<uint8_t*>(tab_blob_.p);
    tab_.fft_len = N; tab_.cp_len = cfg.cp_len; tab_.frame_length = cfg.frame_length; tab_.header_len = h;
    tab_.has_fec = cfg.has_fec ? 1 : 0;
    tab_.n_occ_sets = n_occ_sets; tab_.max_occ = max_occ; tab_.n_pilot_sets = n_pilot_sets; tab_.n_pilots = n_pilots;
    tab_.n_pilot_sym_sets = n_psym; tab_.frame_capacity = capacity_;
    tab_.hdr_bits = cfg.header_bits();   // == the sum over the header symbols' carrier sets (check_header_plan, init())
    tab_.g_min = gmin; tab_.g_max = gmax;
    tab_.eq_alpha = cfg.eq_alpha; tab_.llr_dist_mode = (cfg.llr_dist_mode & 0x1ff) | (cfg.arith == DTL_ARITH_FAST ? DTL_DEMAP_FAST : 0);
    tab_.twiddle = reinterpret_cast<const cf32*>(b + o_tw);
    tab_.bitrev = reinterpret_cast<const uint16_t*>(b + o_br);
    tab_.occ_idx = reinterpret_cast<const int16_t*>(b + o_oi);
    tab_.occ_rank = reinterpret_cast<const int16_t*>(b + o_ork);
    tab_.occ_len = reinterpret_cast<const int16_t*>(b + o_ol);
    tab_.occupied = b + o_oc;
    tab_.pilot_mask = b + o_pm;
    tab_.pilot_val = reinterpret_cast<const cf32*>(b + o_pv);
    tab_.corr_v = reinterpret_cast<const cf32*>(b + o_cv);
    tab_.sync_act = reinterpret_cast<const uint16_t*>(b + o_sa);
    tab_.n_sync_act = int(sync_act.size());
    tab_.ref_sym = reinterpret_cast<const cf32*>(b + o_rf);
    tab_.cnst_pts = reinterpret_cast<const cf32*>(b + o_pt);
    tab_.recip_count = reinterpret_cast<const double*>(b + o_rc);
    for (auto& d : decoders_)
        if (d && !d->uploaded()) { rc = d->upload(); if (rc != DTL_OK) return rc; }
    uploaded_ = true;
    return DTL_OK;
}

// plateau_detector_fb over a continued stream: `edges` (absolute, sorted) extend the carried ones; a run
// that is still above the threshold at `end_abs` is left for the next call unless `final`.
// *produced = every later trigger is >= this position.
static std::vector<int64_t> plateau_stream(std::vector<unsigned long long>& carry, const std::vector<unsigned long long>& edges,
                                           int64_t& scan_i, int64_t end_abs, int max_len, bool final, int64_t* produced)
{
    std::vector<std::pair<int64_t, int64_t>> runs;
    int64_t open = -1;
    auto feed = [&](unsigned long long e) {
        const int64_t pos = int64_t(e >> 1);
        if (e & 1ull) { if (open < 0) open = pos; }
        else if (open >= 0) { runs.emplace_back(open, pos); open = -1; }
    };
    for (unsigned long long e : carry) feed(e);
    for (unsigned long long e : edges) feed(e);
    carry.clear();
    *produced = end_abs;
    if (open >= 0) {
        if (final) runs.emplace_back(open, end_abs);
        else { carry.push_back((static_cast<unsigned long long>(open) << 1) | 1ull); *produced = open; }
    }
    std::vector<int64_t> trig;
    int64_t i = scan_i;
    for (const auto& r : runs) {
        if (r.second <= i) continue; // run entirely behind the scan position
        const int64_t flank_start = std::max(i, r.first);
        i = r.second; // first sample below threshold
        if (i - flank_start > 1) { // "1 sample is not a plateau"
            trig.push_back(flank_start + (i - flank_start) / 2);
            i += max_len;
        }
        i++;
    }
    scan_i = i;
    return trig;
}

static void sort_edges(std::vector<unsigned long long>& edges)
{
    // the kernel appends edges with an atomic counter: sort them (LSD radix, 3 x 11 bits cover 2^33)
    if (edges.size() > 64) {
        std::vector<unsigned long long> tmp(edges.size());
        for (int pass = 0; pass < 3; pass++) {
            size_t cnt[2049] = { 0 };
            const int sh = pass * 11;
            for (unsigned long long e : edges) cnt[((e >> sh) & 2047) + 1]++;
            for (int b = 0; b < 2048; b++) cnt[b + 1] += cnt[b];
            for (unsigned long long e : edges) tmp[cnt[(e >> sh) & 2047]++] = e;
            edges.swap(tmp);
        }
    } else {
        std::sort(edges.begin(), edges.end());
    }
}


int Receiver::process(const cf32* d_iq, int64_t n, unsigned flags, hipStream_t st)
{
    reset_stream();
    return process_stream(d_iq, n, flags, true, st);
}

// per-call context shared by the stages of process_stream (one GR work() call)
struct Receiver::Call {
    const cf32* d_iq = nullptr;     // caller's new samples
    const cf32* iq = nullptr;       // working buffer: [carried tail | new samples]
    int64_t n = 0, W = 0, A = 0, B = 0, E = 0; // new samples [B, E), working buffer [A, E), W = E - A
    unsigned flags = 0;
    bool final = false;
    hipStream_t st = nullptr;
    int N = 0, cp = 0, P = 0, h = 0, L = 0, hdr_syms = 0;
    bool dbg = false;
    double tdbg = 0, t0 = 0;
    int64_t produced = 0;           // frame_detect_bb has been fed the trigger stream up to here
    int n_raw = 0, n_segs = 0, n_cand = 0, n_frames = 0, n_tbs = 0, nofec_stride = 0;
    std::vector<Candidate> cands;
    std::vector<int64_t> cand_abs;
    std::vector<FrameDesc> fdesc;
    std::vector<uint32_t> llr_off;
    uint64_t llr_total = 0;
    std::vector<FrameStat> stats;
    std::vector<int32_t> nofec_info;
    std::vector<TbResult> tb_res;
    uint32_t total_bytes = 0;
    // arg(P) at the raw triggers, submitted group by group as the plateau detector produces them
    struct AnglePart { size_t first, count; };
    std::vector<AnglePart> angle_parts;
    size_t angle_parts_done = 0;  // groups whose angles are on the host and whose oscillator segments are complete
    size_t first_new_seg = 0;     // ss.segs index of this call's first trigger
    // header kernels queued before frame_detect_bb has seen the whole trigger list (speculate_headers): record
    // slot s holds the header found at raw trigger spec_abs[s]; launches in slot order
    struct HdrLaunch { int s0, s1; };
    std::vector<int64_t> spec_abs;
    std::vector<HdrLaunch> hdr_launches;
    size_t spec_raw_next = 0;     // first raw trigger not yet considered
    size_t segs_uploaded = 0;     // ss.segs entries already on the device
    size_t spec_cap = 0;          // record slots the buffers were sized for (0: not sized yet)
    bool spec_on = true;
    bool spec_carry_done = false; // the carried-over triggers have been offered to speculate_headers
    std::unique_ptr<FixSyncRun> fix; // frame_detect_bb over this call's trigger stream, fed range by range
    // the sync kernel's sample ranges of this call (sync_launch): edges of range c at d_edges_ + off, its first `guess`
    // entries copied to the pinned edge buffer together with the range's counter
    struct Range { int64_t mb, me; unsigned max_edges, guess; size_t off; };
    std::vector<Range> rg;
    void lap(const char* what)
    {
        if (!dbg) return;
        const double t = now_ms();
        fprintf(stderr, "  [%s %.3f ms]\n", what, t - tdbg);
        tdbg = t;
    }
};


// the stages were written as one function; they keep its local names through these aliases
#define RX_CALL_LOCALS(c)                                                                                       \
    [[maybe_unused]] StreamState& ss = *ss_;                                                                                   \
    [[maybe_unused]] const cf32* const iq = (c).iq;                                                             \
    [[maybe_unused]] const int64_t n = (c).n, W = (c).W, A = (c).A, B = (c).B, E = (c).E;                       \
    [[maybe_unused]] const unsigned flags = (c).flags;                                                          \
    [[maybe_unused]] const bool final = (c).final, dbg = (c).dbg;                                               \
    [[maybe_unused]] hipStream_t const st = (c).st;                                                             \
    [[maybe_unused]] const int N = (c).N, cp = (c).cp, P = (c).P, h = (c).h, L = (c).L, hdr_syms = (c).hdr_syms; \
    [[maybe_unused]] double& t0 = (c).t0;                                                                       \
    [[maybe_unused]] int64_t& produced = (c).produced;                                                          \
    [[maybe_unused]] int &n_raw = (c).n_raw, &n_segs = (c).n_segs, &n_cand = (c).n_cand, &n_frames = (c).n_frames, \
        &n_tbs = (c).n_tbs, &nofec_stride = (c).nofec_stride;                                                   \
    [[maybe_unused]] auto& cands = (c).cands;                                                                   \
    [[maybe_unused]] auto& cand_abs = (c).cand_abs;                                                             \
    [[maybe_unused]] auto& fdesc = (c).fdesc;                                                                   \
    [[maybe_unused]] auto& llr_off = (c).llr_off;                                                               \
    [[maybe_unused]] uint64_t& llr_total = (c).llr_total;                                                       \
    [[maybe_unused]] auto& stats = (c).stats;                                                                   \
    [[maybe_unused]] auto& nofec_info = (c).nofec_info;                                                         \
    [[maybe_unused]] auto& tb_res = (c).tb_res;                                                                 \
    [[maybe_unused]] uint32_t& total_bytes = (c).total_bytes;                                                   \
    [[maybe_unused]] auto lap = [&](const char* what) { (c).lap(what); }

// working buffer of one call
int Receiver::stream_begin(Call& c)
{
    RX_CALL_LOCALS(c);
    // ---- working buffer: [carried tail | new samples]; A = absolute index of its first sample (a
    // multiple of the sync kernel's tile, so that every metric value is computed exactly as in one batch)
    if (ss.carry_len) {
        DTL_TRY(d_work_.ensure(size_t(W) * sizeof(cf32)));
        DTL_HIP_CHECK(hipMemcpyAsync(d_work_.p, d_carry_.p, size_t(ss.carry_len) * sizeof(cf32), hipMemcpyDeviceToDevice, st));
        if (n) DTL_HIP_CHECK(hipMemcpyAsync(d_work_.as<cf32>() + ss.carry_len, c.d_iq, size_t(n) * sizeof(cf32), hipMemcpyDeviceToDevice, st));
        c.iq = d_work_.as<cf32>();
    }
    lap("work buffer");

    return DTL_OK;
}

// K1, device half: the sync kernel over the call's new samples in up to `sync_ranges_` sample ranges, each with its own
// device -> host copy of the crossings and an event, all queued at once.
int Receiver::sync_launch(Call& c)
{
    RX_CALL_LOCALS(c);
    const int64_t mb0 = B - A;
    // several ranges: what the host does for a range (sort, plateau detector, frame_detect_bb walk, oscillator phases,
    // queueing the range's header kernel) runs while the device is busy with the next range's sync kernel and the
    // header kernels queued so far
    const int n_sc = n >= split_min_ ? sync_ranges_ : 1;
    auto& rg = c.rg;
    rg.clear();
    size_t edge_slots = 0;
    for (int k = 0; k < n_sc && n > 0; k++) {
        int64_t mb = mb0 + n * k / n_sc, me = mb0 + n * (k + 1) / n_sc;
        if (k) mb &= ~int64_t(kSyncTileSamples - 1);               // ranges meet on tile boundaries
        if (k + 1 < n_sc) me &= ~int64_t(kSyncTileSamples - 1);
        if (me <= mb) continue;
        Call::Range r{ mb, me, unsigned(std::min<int64_t>(std::max<int64_t>((me - mb) / 8, 1024), 1 << 26)), 0, edge_slots };
        // crossings expected in a stream of frames: two per preamble; copied back together with the counter
        r.guess = unsigned(std::min<int64_t>(r.max_edges, 4 * ((me - mb) / frame_samples() + 1) + 1024));
        edge_slots += r.max_edges;
        rg.push_back(r);
    }
    if (rg.empty()) return DTL_OK;
    DTL_TRY(d_edges_.ensure(edge_slots * 8));
    DTL_TRY(d_counter_.ensure(64));
    DTL_TRY(h_small_.ensure(128));   // range counters at 0 .. 63, the call's byte total at 96
    size_t guess_total = 0;
    for (auto& r : rg) guess_total += r.guess;
    DTL_TRY(h_edges_.ensure(guess_total * 8));
    DTL_HIP_CHECK(hipMemsetAsync(d_counter_.p, 0, 64, st));
    float* metric = nullptr;
    if (flags & RX_WANT_METRIC) { DTL_TRY(d_metric.ensure(size_t(n) * 4)); metric = d_metric.as<float>(); }
    while (chunk_ev_.size() < rg.size()) {
        hipEvent_t e = nullptr;
        DTL_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        chunk_ev_.push_back(e);
    }
    size_t hoff = 0;
    for (size_t k = 0; k < rg.size(); k++) {
        const Call::Range& r = rg[k];
        KC_BEGIN(st);
        DTL_TRY(launch_sync_edges(iq, r.me, r.mb, N, cfg.sync_threshold, d_edges_.as<unsigned long long>() + r.off,
                                  d_counter_.as<unsigned int>() + k, r.max_edges, metric ? metric + (r.mb - mb0) : nullptr, st));
        KC_END(KernelClock::SYNC, st);
        // (the range's counter and crossings on a stream of their own -- so that two sync kernels follow each other without the
        // two copy dispatches in between -- measured: nothing for plain calls, +1.5 ms per step for batches handed over ahead,
        // whose next sync kernels are already queued when these copies are issued; they stay on the main stream)
        hipStream_t cp = st;
        DTL_HIP_CHECK(hipMemcpyAsync(h_small_.as<unsigned>() + k, d_counter_.as<unsigned int>() + k, 4, hipMemcpyDeviceToHost, cp));
        DTL_HIP_CHECK(hipMemcpyAsync(h_edges_.as<unsigned long long>() + hoff, d_edges_.as<unsigned long long>() + r.off,
                                     size_t(r.guess) * 8, hipMemcpyDeviceToHost, cp));
        DTL_HIP_CHECK(hipEventRecord(chunk_ev_[k], cp));
        hoff += r.guess;
    }
    return DTL_OK;
}

// K1: Schmidl-Cox metric, threshold crossings, plateau detector
int Receiver::stage_sync(Call& c)
{
    RX_CALL_LOCALS(c);
    Call& c_ = c; // the range loops below use `c` as their index
    c.first_new_seg = ss.segs.size();
    // ---- K1: timing metric -> threshold crossings -> plateau detector (host), pipelined over up to four
    // sample ranges: the crossings of one range are sorted and run through the (incremental) plateau
    // detector while the sync kernel works on the next range.
    produced = E;
    double plateau_ms = 0.0;
    unsigned n_edges_total = 0;
    {
        DTL_TRY(sync_launch(c));
        using Range = Call::Range;
        auto& rg = c.rg;
        if (!rg.empty()) {
            lap("launch sync");
            // What the host does per range is two stages of a pipeline: (X) wait for the range's crossings, sort them,
            // run the plateau detector; (Y) frame_detect_bb walk, arg(P) at the triggers, oscillator phases, the range's
            // header kernel.  One thread doing both needs ~0.55 ms per 16 k triggers against 0.34 ms of sync kernel: the
            // device would wait for the host.  With several ranges X runs on a helper thread (it touches the detector's
            // carry state and the pinned edge buffer only; HIP calls: event wait, the rare overflow copy), Y here.
            struct RangeOut { std::vector<int64_t> trig; int64_t produced = 0; unsigned n_edges = 0; int rc = DTL_OK; double ms = 0; bool ready = false; char err[200] = ""; };
            std::vector<RangeOut> outs(rg.size());
            std::atomic<bool> fix_failed{ false };
            std::mutex mu;
            std::condition_variable cv;
            int dev = 0;
            DTL_HIP_CHECK(hipGetDevice(&dev));
            auto stage_x = [&](size_t c, size_t hoff_c, RangeOut& o) -> int {
                const Range& r = rg[c];
                DTL_HIP_CHECK(hipEventSynchronize(chunk_ev_[c]));
                o.n_edges = h_small_.as<unsigned>()[c];
                if (o.n_edges > r.max_edges) { set_last_error("sync: too many threshold crossings (noise-like input?)"); return DTL_ERR_CAPACITY; }
                std::vector<unsigned long long> edges(o.n_edges);
                std::memcpy(edges.data(), h_edges_.as<unsigned long long>() + hoff_c, size_t(std::min(o.n_edges, r.guess)) * 8);
                if (o.n_edges > r.guess) { // more crossings than a frame stream produces: fetch the rest (all ranges are done by then or soon)
                    DTL_HIP_CHECK(hipMemcpy(edges.data() + r.guess, d_edges_.as<unsigned long long>() + r.off + r.guess,
                                            size_t(o.n_edges - r.guess) * 8, hipMemcpyDeviceToHost));
                }
                const double t_e = now_ms();
                sort_edges(edges);
                if (A) for (auto& e : edges) e += static_cast<unsigned long long>(A) << 1; // buffer-relative -> absolute
                o.produced = E;
                o.trig = plateau_stream(ss.edge_carry, edges, ss.plateau_i, A + r.me, cp, final && c + 1 == rg.size(), &o.produced);
                o.ms = now_ms() - t_e;
                return DTL_OK;
            };
            std::vector<size_t> hoffs(rg.size(), 0);
            for (size_t c = 1; c < rg.size(); c++) hoffs[c] = hoffs[c - 1] + rg[c - 1].guess;
            // frame_detect_bb's walk over a range's triggers (what of it can be decided before later triggers and the
            // buffer's length are known; the rest waits for stage_triggers) needs nothing from stage Y and nothing of
            // stage Y needs it: it follows X on the helper thread, after the range's triggers have been handed over
            auto feed_fix = [&](const RangeOut& o) {
                if (!c_.fix) return;
                std::vector<int64_t> rel(o.trig);
                for (auto& t : rel) t -= ss.trig_produced;
                c_.fix->feed(rel.data(), rel.size(), o.produced - ss.trig_produced);
            };
            if (cfg.use_sync_correct) {
                ss.fds.frame_len = (L + hdr_syms) * P; // ofdm_receiver.py:77-78
                c_.fix = std::make_unique<FixSyncRun>(ss.fds);
            }
            // Who touches what while the helper runs: the helper owns ss.edge_carry, ss.plateau_i, ss.fds (through c_.fix)
            // and reads the pinned edge / counter buffers; this thread owns everything else of the stream state (ss.segs,
            // the trigger lists, the candidate bookkeeping).  RangeOut entries cross over under `mu` (`ready`).
            // `cancel`: set before this thread leaves the stage early (an error), so that the helper stops at the next range
            // instead of walking all of them; the failed call's stream state is discarded anyway (process_stream resets it).
            std::atomic<bool> cancel{ false };
            auto helper_job = [&]() {
                (void)hipSetDevice(dev);                          // (a new thread starts on device 0)
                int fail_rc = DTL_OK;                             // != OK: this and every later range report it
                const char* fail_text = "";
                for (size_t c = 0; c < rg.size(); c++) {
                    RangeOut& o = outs[c];
                    if (fail_rc == DTL_OK && cancel.load(std::memory_order_acquire)) { fail_rc = DTL_ERR_HIP; fail_text = "sync stage: cancelled"; }
                    if (fail_rc == DTL_OK) {
                        try {
                            o.rc = stage_x(c, hoffs[c], o);
                            if (o.rc != DTL_OK) { fail_rc = o.rc; fail_text = last_error(); /* the error text is per thread */ }
                        } catch (const std::bad_alloc&) {
                            fail_rc = DTL_ERR_CAPACITY; fail_text = "sync stage: out of host memory";
                        } catch (...) {
                            fail_rc = DTL_ERR_INTERNAL; fail_text = "sync stage: unexpected exception";
                        }
                    }
                    if (fail_rc != DTL_OK) { o.rc = fail_rc; std::snprintf(o.err, sizeof(o.err), "%s", fail_text); }
                    { std::lock_guard<std::mutex> lk(mu); o.ready = true; }
                    cv.notify_all();
                    if (fail_rc == DTL_OK && !cancel.load(std::memory_order_acquire)) {
                        try {
                            const double t_f = now_ms(); feed_fix(o); o.ms += now_ms() - t_f;
                        } catch (...) {                           // the walk's vectors could not grow
                            fail_rc = DTL_ERR_CAPACITY; fail_text = "frame_detect walk: out of host memory";
                            fix_failed.store(true, std::memory_order_release);
                        }
                    }
                }
            };
            bool helper_on = false;
            if (rg.size() > 1 && !worker_failed_) {
                // a helper that cannot be started (std::system_error from std::thread, std::bad_alloc) is no reason to fail
                // the call: the stage then runs on this thread alone, and the handle does not try again
                try {
                    if (getenv("DTL_RX_FAIL_HELPER")) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "DTL_RX_FAIL_HELPER"); // (tests)
                    if (!worker_) worker_ = std::make_unique<Worker>();
                    worker_->run(helper_job);
                    helper_on = true;
                } catch (const std::exception&) {
                    worker_failed_ = true;
                    worker_.reset();
                }
            }
            // declared behind everything the helper's job refers to: destroyed first, i.e. the job has ended before any of it
            // goes out of scope, whichever way this function is left
            struct Joiner {
                Worker* w; std::atomic<bool>& cancel;
                ~Joiner() { if (w) { cancel.store(true, std::memory_order_release); w->wait(); } }
            } joiner{ helper_on ? worker_.get() : nullptr, cancel };
            for (size_t c = 0; c < rg.size(); c++) {
                const bool last = c + 1 == rg.size();
                RangeOut& o = outs[c];
                if (helper_on) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return o.ready; });
                } else {
                    o.rc = stage_x(c, hoffs[c], o);
                }
                lap("  range: crossings sorted, plateaus found");
                if (o.rc != DTL_OK) { if (o.err[0]) set_last_error(o.err); return o.rc; }
                const unsigned n_edges = o.n_edges;
                std::vector<int64_t>& trig = o.trig;
                produced = o.produced;
                const size_t first = raw_triggers.size();
                raw_triggers.insert(raw_triggers.end(), trig.begin(), trig.end());
                n_edges_total += n_edges;
                // K1b for this range's triggers on the side stream (the working buffer is complete: the event waited
                // for sits behind its copies) ...
                if (!side_) {
                    int least = 0, greatest = 0;
                    DTL_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
                    DTL_HIP_CHECK(hipStreamCreateWithPriority(&side_, hipStreamNonBlocking, greatest));
                }
                const size_t parts_before = c_.angle_parts.size();   // (a range without triggers submits nothing)
                DTL_TRY(submit_angles(c_, first, trig.size(), side_));
                lap("  range: submit angles");
                if (!helper_on) { feed_fix(o); lap("  range: frame_detect"); }  // (no helper: everything on this thread)
                // The oscillator segments, then the headers of every trigger whose segments are complete and whose header
                // symbols lie in front of `produced` (a later range decides what the others see): queued behind the sync
                // kernels, they keep the device busy while the host deals with the next range -- and, for the last one, with
                // frame_detect_bb's rest and the candidate list (stage_triggers).  Skewed by one range: the angles of range
                // c - 1 are taken when range c's arg(P) pass has been queued -- they have arrived by then, where waiting for
                // them right behind their own submission cost a device round trip (0.27 ms) per range with the device idle.
                // A call of a single range takes the unpipelined path (stage_frames launches everything).
                if (c > 0 && parts_before > 0) {
                    DTL_TRY(collect_angles(c_, parts_before));   // ranges 0 .. c-1
                    lap("  range: collect angles (previous range)");
                    DTL_TRY(speculate_headers(c_, produced, false));
                    lap("  range: speculate headers (previous range)");
                }
                if (last && c_.spec_on && !c_.hdr_launches.empty()) {
                    DTL_TRY(collect_angles(c_, c_.angle_parts.size()));
                    lap("  range: collect angles");
                    DTL_TRY(speculate_headers(c_, final ? std::numeric_limits<int64_t>::max() : produced, true));
                    lap("  range: speculate headers");
                }
            }
            if (helper_on) worker_->wait();
            if (fix_failed.load(std::memory_order_acquire)) { set_last_error("frame_detect walk: out of host memory"); return DTL_ERR_CAPACITY; }
            for (const auto& o : outs) plateau_ms += o.ms;
            lap("helper joined");
        } else {
            // no new samples (a final flush): close what is open
            raw_triggers = plateau_stream(ss.edge_carry, {}, ss.plateau_i, E, cp, final, &produced);
            DTL_TRY(submit_angles(c_, 0, raw_triggers.size(), st));
        }
    }
    dbg_n_edges = n_edges_total;
    stage_ms[0] = now_ms() - t0; t0 = now_ms();
    dbg_plateau_ms = plateau_ms;

    return DTL_OK;
}

// K1b for raw_triggers[first, first + count): positions up, arg(P) kernel, angles down -- all on `on`, closed by
// an event; the oscillator segments (sample_and_hold + frequency_modulator_fc) are opened here and completed by
// collect_angles once the angles are on the host
int Receiver::submit_angles(Call& c, size_t first, size_t count, hipStream_t on)
{
    StreamState& ss = *ss_;
    ss.segs.reserve(ss.segs.size() + count);
    for (size_t j = 0; j < count; j++) {
        CfoSegment sg{};
        sg.start = raw_triggers[first + j];
        ss.segs.push_back(sg);
    }
    trigger_angles.resize(first + count, 0.0f);
    if (!count) return DTL_OK;
    const size_t need = first + count;
    if (need * 8 > h_trig_.cap || need * 8 > d_trig_.cap || need * 4 > d_angle_.cap || need * 4 > h_angle_.cap) {
        // growing the buffers discards them: take what is in flight first
        DTL_TRY(collect_angles(c, c.angle_parts.size()));
        const size_t want = std::max<size_t>(2 * need, size_t(2 * (c.W / frame_samples() + 1) + 1024));
        h_trig_.device_reads = zero_copy_trig_;
        DTL_TRY(h_trig_.ensure(want * 8));
        DTL_TRY(d_trig_.ensure(want * 8));
        DTL_TRY(d_angle_.ensure(want * 4));
        DTL_TRY(h_angle_.ensure(want * 4));
    }
    int64_t* rel = h_trig_.as<int64_t>() + first;
    for (size_t j = 0; j < count; j++) rel[j] = raw_triggers[first + j] - c.A;
    const int64_t* trig_dev = d_trig_.as<int64_t>() + first;
    if (zero_copy_trig_) {                                // one 8-byte read per trigger, in place
        void* dp = nullptr;
        DTL_TRY(device_view(h_trig_, &dp));
        trig_dev = static_cast<const int64_t*>(dp) + first;
    } else {
        DTL_HIP_CHECK(hipMemcpyAsync(d_trig_.as<int64_t>() + first, rel, count * 8, hipMemcpyHostToDevice, on));
    }
    DTL_TRY(launch_trigger_angle(c.iq, c.W, c.N, trig_dev, int(count), d_angle_.as<float>() + first, on));
    DTL_HIP_CHECK(hipMemcpyAsync(h_angle_.as<float>() + first, d_angle_.as<float>() + first, count * 4, hipMemcpyDeviceToHost, on));
    const size_t part = c.angle_parts.size();
    while (angle_ev_.size() <= part) {
        hipEvent_t e = nullptr;
        DTL_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        angle_ev_.push_back(e);
    }
    DTL_HIP_CHECK(hipEventRecord(angle_ev_[part], on));
    c.angle_parts.push_back({ first, count });
    return DTL_OK;
}

// wait for the first `parts` groups and finish their oscillator segments (in order: each segment starts with the
// previous segment's phase advanced to the new trigger; before the first trigger the held angle is 0 and the
// phase stays 0)
int Receiver::collect_angles(Call& c, size_t parts)
{
    StreamState& ss = *ss_;
    const float sens = float(-2.0 / c.N);
    for (; c.angle_parts_done < parts; c.angle_parts_done++) {
        const Call::AnglePart& ap = c.angle_parts[c.angle_parts_done];
        DTL_HIP_CHECK(hipEventSynchronize(angle_ev_[c.angle_parts_done]));
        std::memcpy(trigger_angles.data() + ap.first, h_angle_.as<float>() + ap.first, ap.count * 4);
        for (size_t j = ap.first; j < ap.first + ap.count; j++) {
            CfoSegment& sg = ss.segs[c.first_new_seg + j];
            sg.inc = osc_phase_step(sens * trigger_angles[j]);
            sg.phase0 = 0u;
            if (c.first_new_seg + j > 0) {
                const CfoSegment& pv = ss.segs[c.first_new_seg + j - 1];
                sg.phase0 = pv.phase0 + pv.inc * uint32_t(sg.start - pv.start);     // (mod 2^32: the wrap)
            }
        }
    }
    return DTL_OK;
}

// Phase A ahead of time.  frame_detect_bb needs the whole trigger list of a call, but it leaves nearly every raw
// trigger where it is; so the header kernel is queued for raw triggers as soon as everything it reads is final:
// the trigger's own oscillator segment and every segment that starts inside its header symbols (`limit` = the
// position up to which the raw trigger list is final, cut back to the first segment still waiting for its angle).  stage_frames maps the fixed triggers onto these record
// slots and runs the kernel only for the ones that are missing (moved or inserted triggers, the last range).
int Receiver::speculate_headers(Call& c, int64_t limit, bool split_last)
{
    StreamState& ss = *ss_;
    if (!c.spec_on || c.angle_parts_done == 0) return DTL_OK;
    const int N = c.N, P = c.P;
    const Call::AnglePart& lastp = c.angle_parts[c.angle_parts_done - 1];
    const size_t done_raw = lastp.first + lastp.count;          // raw triggers whose segments are complete
    const int64_t need = int64_t(c.final ? c.hdr_syms : c.hdr_syms + c.L) * P;
    const int64_t span = int64_t(c.hdr_syms + 1) * P;
    // first position whose oscillator state is not final: the first trigger still waiting for its angle, else the
    // point up to which the trigger list itself is final
    if (done_raw < raw_triggers.size()) limit = std::min(limit, raw_triggers[done_raw]);
    std::vector<Candidate> add;
    std::vector<int64_t> add_abs;
    if (!c.spec_carry_done) {
        // triggers carried over from the previous call (their frames were incomplete then) come first in the walk
        c.spec_carry_done = true;
        for (int64_t tpos : ss.pending_fixed) {
            if (tpos + need > c.E || tpos + span > limit) break;
            size_t ri = 0;
            while (ri < ss.segs.size() && ss.segs[ri].start <= tpos) ri++;
            if (ri > c.first_new_seg + done_raw) break;         // would read a segment that is not complete yet
            Candidate cd{};
            cd.trigger = tpos - c.A;
            cd.seg = int(ri) - 1;
            add.push_back(cd);
            add_abs.push_back(tpos);
        }
    }
    size_t j = c.spec_raw_next;
    for (; j < done_raw; j++) {
        const int64_t tpos = raw_triggers[j];
        if (tpos + span > limit) break;                         // a later range decides what this header sees
        if (tpos + need > c.E) continue;                        // never a candidate of this call
        Candidate cd{};
        cd.trigger = tpos - c.A;
        cd.seg = int(c.first_new_seg + j);
        add.push_back(cd);
        add_abs.push_back(tpos);
    }
    c.spec_raw_next = j;
    if (add.empty()) return DTL_OK;
    if (!c.spec_cap) { // size everything the early launches write for twice the frames the call can hold
        const size_t cap = size_t(2 * (c.W / frame_samples() + 1) + 1024);
        const size_t seg_cap = ss.segs.size() + cap;
        DTL_TRY(d_cands_.ensure(cap * sizeof(Candidate)));
        h_cands_.device_reads = zero_copy_cands_;
        DTL_TRY(h_cands_.ensure(cap * sizeof(Candidate)));
        DTL_TRY(d_segs_.ensure(seg_cap * sizeof(CfoSegment)));
        DTL_TRY(h_segs_.ensure(seg_cap * sizeof(CfoSegment)));
        DTL_TRY(d_recs_.ensure(cap * sizeof(HeaderRec)));
        DTL_TRY(h_recs_.ensure(cap * sizeof(HeaderRec)));
        DTL_TRY(d_taps.ensure(cap * N * sizeof(cf32)));
        if (c.flags & RX_WANT_HDR_BITS) DTL_TRY(d_hdr_bits.ensure(cap * tab_.hdr_bits));
        c.spec_cap = cap;
    }
    const size_t s0 = c.spec_abs.size(), s1 = s0 + add.size();
    const size_t segs_now = c.first_new_seg + done_raw;         // complete segments
    if (s1 > c.spec_cap || segs_now * sizeof(CfoSegment) > d_segs_.cap || segs_now * sizeof(CfoSegment) > h_segs_.cap) {
        c.spec_on = false;                                      // more triggers than a stream of frames produces
        return DTL_OK;
    }
    std::memcpy(h_cands_.as<Candidate>() + s0, add.data(), add.size() * sizeof(Candidate));
    const Candidate* cands_dev = d_cands_.as<Candidate>();
    if (zero_copy_cands_) {                               // one 16-byte read per header workgroup, in place
        void* dp = nullptr;
        DTL_TRY(device_view(h_cands_, &dp));
        cands_dev = static_cast<const Candidate*>(dp);
    } else {
        DTL_HIP_CHECK(hipMemcpyAsync(d_cands_.as<Candidate>() + s0, h_cands_.as<Candidate>() + s0, add.size() * sizeof(Candidate),
                                     hipMemcpyHostToDevice, c.st));
    }
    if (segs_now > c.segs_uploaded) {
        CfoSegment* hs = h_segs_.as<CfoSegment>();
        for (size_t k = c.segs_uploaded; k < segs_now; k++) { hs[k] = ss.segs[k]; hs[k].start -= c.A; }
        DTL_HIP_CHECK(hipMemcpyAsync(d_segs_.as<CfoSegment>() + c.segs_uploaded, hs + c.segs_uploaded,
                                     (segs_now - c.segs_uploaded) * sizeof(CfoSegment), hipMemcpyHostToDevice, c.st));
        c.segs_uploaded = segs_now;
    }
    uint8_t* hb = (c.flags & RX_WANT_HDR_BITS) ? d_hdr_bits.as<uint8_t>() : nullptr;
    // the call's last group goes out in four launches, each with its own record copy and event: the demux walk of one
    // part (stage_frames) then overlaps the kernels of the following ones (earlier groups are long done by then)
    const int n_add = int(add.size()), n_parts = (split_last && n_add >= 8192) ? 4 : 1;
    for (int part = 0; part < n_parts; part++) {
        const int e0 = int(int64_t(n_add) * part / n_parts), e1 = int(int64_t(n_add) * (part + 1) / n_parts);
        if (e1 <= e0) continue;
        const size_t p0 = s0 + size_t(e0);
        KC_BEGIN(c.st);
        DTL_TRY(launch_header(tab_, c.iq, c.W, cands_dev + p0, e1 - e0, d_segs_.as<CfoSegment>(),
                              int(c.segs_uploaded), d_recs_.as<HeaderRec>() + p0, d_taps.as<cf32>() + p0 * N,
                              hb ? hb + p0 * tab_.hdr_bits : nullptr, c.st));
        KC_END(KernelClock::HEADER, c.st);
        DTL_HIP_CHECK(hipMemcpyAsync(h_recs_.as<HeaderRec>() + p0, d_recs_.as<HeaderRec>() + p0, size_t(e1 - e0) * sizeof(HeaderRec),
                                     hipMemcpyDeviceToHost, c.st));
        while (hdr_ev_.size() <= c.hdr_launches.size()) {
            hipEvent_t e = nullptr;
            DTL_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            hdr_ev_.push_back(e);
        }
        DTL_HIP_CHECK(hipEventRecord(hdr_ev_[c.hdr_launches.size()], c.st));
        c.hdr_launches.push_back({ int(p0), int(p0) + (e1 - e0) });
    }
    c.spec_abs.insert(c.spec_abs.end(), add_abs.begin(), add_abs.end());
    return DTL_OK;
}

// K1b + frame_detect_bb + candidate list
int Receiver::stage_triggers(Call& c)
{
    RX_CALL_LOCALS(c);
    // ---- K1b (arg(P) at the raw triggers) was submitted range by range in stage_sync; frame_detect_bb and the
    // candidate list only need the trigger POSITIONS and run while the last group is in flight
    n_raw = int(raw_triggers.size());
    n_segs = int(ss.segs.size());
    lap("sync ranges");
    // ---- frame_detect_bb (optional): one work() buffer = the trigger stream produced by this call
    std::vector<int64_t> new_fixed;
    if (cfg.use_sync_correct) {
        ss.fds.frame_len = (L + hdr_syms) * P; // ofdm_receiver.py:77-78
        const int64_t c0 = ss.trig_produced, len = produced - c0;
        if (len > 0) {
            if (!c.fix) { // nothing was fed during the sync stage (a flush without new samples)
                std::vector<int64_t> rel(raw_triggers);
                for (auto& t : rel) t -= c0;
                c.fix = std::make_unique<FixSyncRun>(ss.fds);
                c.fix->feed(rel.data(), rel.size(), 0);
            }
            new_fixed = c.fix->finish(len);
            for (auto& t : new_fixed) t += c0;
        }
    } else {
        new_fixed = raw_triggers;
    }
    lap("fix_sync");
    ss.trig_produced = produced;
    fixed_triggers = new_fixed;
    ss.pending_fixed.insert(ss.pending_fixed.end(), new_fixed.begin(), new_fixed.end());
    // candidates for phase A: every pending trigger whose samples are here (a continued stream waits
    // for the whole frame, the end of a stream only needs the header symbols)
    cands.clear();
    cand_abs.clear();
    cands.reserve(ss.pending_fixed.size());
    cand_abs.reserve(ss.pending_fixed.size());
    {
        size_t ri = 0; // number of segments starting <= tpos (both lists are sorted)
        const int64_t need = int64_t(final ? hdr_syms : hdr_syms + L) * P;
        // ... and the raw trigger list must be final up to the frame's last sample: a trigger found by a later call
        // (its plateau is still open at the end of this one) starts a new oscillator segment inside the frame
        const int64_t ready = final ? E : std::min(E, produced);
        for (int64_t tpos : ss.pending_fixed) {
            if (tpos + need > ready) break;
            while (ri < ss.segs.size() && ss.segs[ri].start <= tpos) ri++;
            Candidate c{};
            c.trigger = tpos - A;
            c.seg = int(ri) - 1;
            cands.push_back(c);
            cand_abs.push_back(tpos);
        }
    }
    n_cand = int(cands.size());
    lap("candidates");
    DTL_TRY(collect_angles(c, c.angle_parts.size()));
    lap("segment phases");
    stage_ms[1] = now_ms() - t0; t0 = now_ms();

    return DTL_OK;
}

// ------------------------------------------------------------------ single blocks (frequency-domain vectors in)

// the per-block entry points accept NaN, zeros and normal floats up to 2^62 in magnitude (see range_check_kernel)
static int check_block_inputs(const char* what, std::initializer_list<std::pair<const cf32*, size_t>> arrays, hipStream_t st)
{
    DevBuf flag;
    DTL_TRY(flag.ensure(16));
    DTL_HIP_CHECK(hipMemsetAsync(flag.p, 0, 4, st));
    for (const auto& a : arrays)
        if (a.first && a.second) DTL_TRY(launch_range_check(reinterpret_cast<const float*>(a.first), int64_t(a.second) * 2, flag.as<unsigned int>(), st));
    unsigned int bad = 0;
    DTL_HIP_CHECK(hipMemcpyAsync(&bad, flag.p, 4, hipMemcpyDeviceToHost, st));
    DTL_HIP_CHECK(hipStreamSynchronize(st));
    if (bad) {
        set_last_error(std::string(what) + ": input outside the supported range (every component NaN, 0 or a normal float with "
                                           "|x| <= 2^62: denormals, infinities and larger values do not give the reference's result)");
        return DTL_ERR_INVALID_ARG;
    }
    return DTL_OK;
}

int Receiver::chanest(const cf32* d_sync_syms, int n, int32_t* d_carr_off, cf32* d_taps, hipStream_t st)
{
    if (n < 0 || (n > 0 && (!d_sync_syms || !d_carr_off || !d_taps))) { set_last_error("chanest: bad argument"); return DTL_ERR_INVALID_ARG; }
    DeviceScope dev;
    DTL_TRY(dev.enter(cfg.device));
    if (!uploaded_) DTL_TRY(upload_tables());
    DTL_TRY(check_block_inputs("chanest", { { d_sync_syms, size_t(n) * 2 * cfg.fft_len } }, st));
    HeaderStageIo io;
    io.freq_in = d_sync_syms; io.n_rows = 2; io.row0 = 0;
    io.taps_raw_out = d_taps; io.carr_off_out = d_carr_off; io.do_header = 0;
    return launch_header_stage(tab_, n, io, nullptr, nullptr, nullptr, st);
}

int Receiver::header_decode(const cf32* d_hdr_syms, const cf32* d_taps, const int32_t* d_carr_off, int n,
                            HeaderRec* recs_host, uint8_t* d_hdr_bits, cf32* d_taps_out, hipStream_t st)
{
    if (n < 0 || (n > 0 && (!d_hdr_syms || !d_taps || !d_carr_off || !recs_host))) { set_last_error("header_decode: bad argument"); return DTL_ERR_INVALID_ARG; }
    if (n == 0) return DTL_OK;
    DeviceScope dev;
    DTL_TRY(dev.enter(cfg.device));
    if (!uploaded_) DTL_TRY(upload_tables());
    DTL_TRY(check_block_inputs("header_decode", { { d_hdr_syms, size_t(n) * cfg.header_len() * cfg.fft_len }, { d_taps, size_t(n) * cfg.fft_len } }, st));
    DevBuf recs, taps_scratch;
    DTL_TRY(recs.ensure(size_t(n) * sizeof(HeaderRec)));
    if (!d_taps_out) { DTL_TRY(taps_scratch.ensure(size_t(n) * cfg.fft_len * sizeof(cf32))); d_taps_out = taps_scratch.as<cf32>(); }
    HeaderStageIo io;
    io.freq_in = d_hdr_syms; io.n_rows = cfg.header_len(); io.row0 = 2;
    io.taps_in = d_taps; io.carr_off_in = d_carr_off; io.do_header = 1;
    DTL_TRY(launch_header_stage(tab_, n, io, recs.as<HeaderRec>(), d_taps_out, d_hdr_bits, st));
    DTL_HIP_CHECK(hipMemcpyAsync(recs_host, recs.p, size_t(n) * sizeof(HeaderRec), hipMemcpyDeviceToHost, st));
    DTL_HIP_CHECK(hipStreamSynchronize(st));
    return DTL_OK;
}

int Receiver::equalize_frames(const cf32* d_frames, const cf32* d_taps, const FrameDesc* descs_host, int n, cf32* d_hard,
                              cf32* d_soft, cf32* d_soft_syms, FrameStat* stats_host, hipStream_t st)
{
    if (n < 0 || (n > 0 && (!d_frames || !d_taps || !descs_host))) { set_last_error("equalize_frames: bad argument"); return DTL_ERR_INVALID_ARG; }
    if (n == 0) return DTL_OK;
    for (int i = 0; i < n; i++)
        if (descs_host[i].cnst < 1 || descs_host[i].cnst > 4) { // frame_equalizer_vcvc_impl.cc:148-150
            set_last_error("Missing constellation tag.");
            return DTL_ERR_TAGS_MISSING;
        }
    DeviceScope dev;
    DTL_TRY(dev.enter(cfg.device));
    if (!uploaded_) DTL_TRY(upload_tables());
    DTL_TRY(check_block_inputs("equalize_frames", { { d_frames, size_t(n) * cfg.frame_length * cfg.fft_len }, { d_taps, size_t(n) * cfg.fft_len } }, st));
    DevBuf descs, stats, syms_scratch;
    DTL_TRY(descs.ensure(size_t(n) * sizeof(FrameDesc)));
    DTL_TRY(stats.ensure(size_t(n) * sizeof(FrameStat)));
    if (!d_soft_syms) { DTL_TRY(syms_scratch.ensure(size_t(n) * capacity_ * sizeof(cf32))); d_soft_syms = syms_scratch.as<cf32>(); }
    else DTL_HIP_CHECK(hipMemsetAsync(d_soft_syms, 0, size_t(n) * capacity_ * sizeof(cf32), st));
    DTL_HIP_CHECK(hipMemcpyAsync(descs.p, descs_host, size_t(n) * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
    DTL_TRY(launch_frame_eq_stage(tab_, d_frames, descs.as<FrameDesc>(), n, d_taps, stats.as<FrameStat>(), d_hard, d_soft,
                                  d_soft_syms, st));
    if (stats_host) DTL_HIP_CHECK(hipMemcpyAsync(stats_host, stats.p, size_t(n) * sizeof(FrameStat), hipMemcpyDeviceToHost, st));
    DTL_HIP_CHECK(hipStreamSynchronize(st));
    return DTL_OK;
}

// phase A (headers, demux walk) and phase B (payload kernels)
int Receiver::stage_frames(Call& c)
{
    RX_CALL_LOCALS(c);
    // ---- phases A and B, pipelined over header-kernel launches.  Most candidates already have a record slot from a
    // launch made during the sync stage (speculate_headers); the rest (triggers frame_detect_bb moved or inserted,
    // the triggers of the last sample range, everything in a short call) get theirs now.  While the GPU works
    // through the launches the host takes launch after launch of header records (header_payload_demux walk +
    // header_parser state, both sequential) and queues the payload kernels of the frames found so far behind them.
    fdesc.clear();
    llr_off.clear();
    fdesc.reserve(n_cand);
    llr_off.reserve(n_cand);
    frames.reserve(n_cand);
    const int max_bps = 4;
    nofec_stride = ((capacity_ * 4 / 8 + 8) + 15) & ~15;
    cf32 *eh = nullptr, *es = nullptr, *ss_out = nullptr;
    uint8_t* nofec_bytes = nullptr;
    int32_t* nofec_inf = nullptr;
    llr_total = uint64_t(ss.llr_carry_len);
    std::vector<int> slot(size_t(n_cand), -1);       // record slot of candidate ci
    std::vector<HeaderRec> recs;
    if (n_cand) {
        // slots of the early launches, then the missing candidates in order
        std::vector<int> extra;
        {
            size_t sp = 0;
            for (int ci = 0; ci < n_cand; ci++) {
                while (sp < c.spec_abs.size() && c.spec_abs[sp] < cand_abs[ci]) sp++;
                if (sp < c.spec_abs.size() && c.spec_abs[sp] == cand_abs[ci]) slot[ci] = int(sp);
                else extra.push_back(ci);
            }
        }
        size_t n_slots = c.spec_abs.size() + extra.size();
        const size_t n_segs_all = ss.segs.size();
        if (!c.hdr_launches.empty() &&
            (n_slots > c.spec_cap || n_segs_all * sizeof(CfoSegment) > d_segs_.cap || n_segs_all * sizeof(CfoSegment) > h_segs_.cap)) {
            // the buffers of the early launches cannot take the rest: let them finish and start over without them
            DTL_HIP_CHECK(hipStreamSynchronize(st));
            c.hdr_launches.clear();
            c.spec_abs.clear();
            c.segs_uploaded = 0;
            extra.resize(size_t(n_cand));
            for (int ci = 0; ci < n_cand; ci++) { slot[ci] = -1; extra[ci] = ci; }
            n_slots = extra.size();
        }
        const size_t s_base = c.spec_abs.size();
        for (size_t e = 0; e < extra.size(); e++) slot[extra[e]] = int(s_base + e);
        recs.resize(n_slots);
        DTL_TRY(d_cands_.ensure(n_slots * sizeof(Candidate)));
        h_cands_.device_reads = zero_copy_cands_;
        DTL_TRY(h_cands_.ensure(n_slots * sizeof(Candidate)));
        DTL_TRY(d_segs_.ensure(std::max<size_t>(1, n_segs_all) * sizeof(CfoSegment)));
        DTL_TRY(h_segs_.ensure(std::max<size_t>(1, n_segs_all) * sizeof(CfoSegment)));
        DTL_TRY(d_recs_.ensure(n_slots * sizeof(HeaderRec)));
        DTL_TRY(h_recs_.ensure(n_slots * sizeof(HeaderRec)));
        DTL_TRY(d_taps.ensure(n_slots * N * sizeof(cf32)));
        uint8_t* hb = nullptr;
        if (flags & RX_WANT_HDR_BITS) { DTL_TRY(d_hdr_bits.ensure(n_slots * tab_.hdr_bits)); hb = d_hdr_bits.as<uint8_t>(); }
        // per-frame buffers of phase B, sized for "every candidate becomes a frame at the densest constellation"
        const uint64_t llr_bound = llr_total + uint64_t(n_cand) * capacity_ * max_bps;
        DTL_TRY(d_llr.ensure((size_t(std::min<uint64_t>(llr_bound, 0xFFFFFFF0ull)) + 4096) * 4));
        if (ss.llr_carry_len)
            DTL_HIP_CHECK(hipMemcpyAsync(d_llr.p, d_llr_carry_.p, size_t(ss.llr_carry_len) * 4, hipMemcpyDeviceToDevice, st));
        DTL_TRY(d_stats_.ensure(size_t(n_cand) * sizeof(FrameStat)));
        DTL_TRY(d_frames_.ensure(size_t(n_cand) * sizeof(FrameDesc)));
        h_frames_.device_reads = zero_copy_descs_;
        DTL_TRY(h_frames_.ensure(size_t(n_cand) * sizeof(FrameDesc)));
        if (flags & RX_WANT_EQ_HARD) { DTL_TRY(d_eq_hard.ensure(size_t(n_cand) * L * N * sizeof(cf32))); eh = d_eq_hard.as<cf32>(); }
        if (flags & RX_WANT_EQ_SOFT) { DTL_TRY(d_eq_soft.ensure(size_t(n_cand) * L * N * sizeof(cf32))); es = d_eq_soft.as<cf32>(); }
        // the equalised data symbols cross HBM (frame equaliser -> demapper) only in the two-kernel form of phase B; the
        // fused form writes them when the caller asks for the serializer's stream.  The buffer is cleared only when the
        // caller looks at it (positions behind a frame's last symbol)
        if ((flags & RX_WANT_SOFT_SYMS) || !payload_is_fused(tab_, eh || es)) {
            DTL_TRY(d_soft_syms.ensure(size_t(n_cand) * capacity_ * sizeof(cf32)));
            if (flags & RX_WANT_SOFT_SYMS) DTL_HIP_CHECK(hipMemsetAsync(d_soft_syms.p, 0, size_t(n_cand) * capacity_ * sizeof(cf32), st));
            ss_out = d_soft_syms.as<cf32>();
        }
        if (!cfg.has_fec) {
            DTL_TRY(d_staging_.ensure(size_t(n_cand) * nofec_stride));
            DTL_TRY(d_nofec_info_.ensure(size_t(n_cand) * 8));
            DTL_HIP_CHECK(hipMemsetAsync(d_nofec_info_.p, 0, size_t(n_cand) * 8, st));
            nofec_bytes = d_staging_.as<uint8_t>();
            nofec_inf = d_nofec_info_.as<int32_t>();
        }
        // the remaining oscillator segments, then the missing candidates in up to four launches
        if (n_segs_all > c.segs_uploaded) {
            CfoSegment* hs = h_segs_.as<CfoSegment>();
            for (size_t k = c.segs_uploaded; k < n_segs_all; k++) { hs[k] = ss.segs[k]; hs[k].start -= A; }
            DTL_HIP_CHECK(hipMemcpyAsync(d_segs_.as<CfoSegment>() + c.segs_uploaded, hs + c.segs_uploaded,
                                         (n_segs_all - c.segs_uploaded) * sizeof(CfoSegment), hipMemcpyHostToDevice, st));
            c.segs_uploaded = n_segs_all;
        }
        if (!extra.empty()) {
            Candidate* hc = h_cands_.as<Candidate>() + s_base;
            for (size_t e = 0; e < extra.size(); e++) hc[e] = cands[extra[e]];
            const Candidate* cands_dev = d_cands_.as<Candidate>();
            if (zero_copy_cands_) {
                void* dp = nullptr;
                DTL_TRY(device_view(h_cands_, &dp));
                cands_dev = static_cast<const Candidate*>(dp);
            } else {
                DTL_HIP_CHECK(hipMemcpyAsync(d_cands_.as<Candidate>() + s_base, hc, extra.size() * sizeof(Candidate), hipMemcpyHostToDevice, st));
            }
            const int n_extra = int(extra.size()), n_parts = n_extra >= 8192 ? 4 : 1;
            for (int part = 0; part < n_parts; part++) {
                const int e0 = int(int64_t(n_extra) * part / n_parts), e1 = int(int64_t(n_extra) * (part + 1) / n_parts);
                if (e1 <= e0) continue;
                const size_t s0 = s_base + size_t(e0);
                KC_BEGIN(st);
                DTL_TRY(launch_header(tab_, iq, W, cands_dev + s0, e1 - e0, d_segs_.as<CfoSegment>(), n_segs,
                                      d_recs_.as<HeaderRec>() + s0, d_taps.as<cf32>() + s0 * N,
                                      hb ? hb + s0 * tab_.hdr_bits : nullptr, st));
                KC_END(KernelClock::HEADER, st);
                DTL_HIP_CHECK(hipMemcpyAsync(h_recs_.as<HeaderRec>() + s0, d_recs_.as<HeaderRec>() + s0,
                                             size_t(e1 - e0) * sizeof(HeaderRec), hipMemcpyDeviceToHost, st));
                while (hdr_ev_.size() <= c.hdr_launches.size()) {
                    hipEvent_t e = nullptr;
                    DTL_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    hdr_ev_.push_back(e);
                }
                DTL_HIP_CHECK(hipEventRecord(hdr_ev_[c.hdr_launches.size()], st));
                c.hdr_launches.push_back({ int(s0), int(s0) + (e1 - e0) });
            }
        }
    }
    int n_walked = 0; // candidates the demux has dealt with
    bool stop = false, payload_started = false;
    size_t ready = 0; // header launches whose records are on the host
    int f0 = 0;       // frames already handed to phase B
    // phase B for the frames found so far (tb bookkeeping runs on the host while it executes)
    auto flush_payload = [&]() -> int {
        const int f1 = int(fdesc.size());
        if (f1 <= f0) return DTL_OK;
        // the descriptors stay where the host wrote them: the frame equaliser reads them there (zero_copy_descs_) and
        // leaves the device copy the demapper reads
        std::memcpy(h_frames_.as<FrameDesc>() + f0, fdesc.data() + f0, size_t(f1 - f0) * sizeof(FrameDesc));
        const FrameDesc* host_descs = nullptr;
        if (zero_copy_descs_) {
            void* dp = nullptr;
            DTL_HIP_CHECK(hipHostGetDevicePointer(&dp, h_frames_.p, 0));
            host_descs = static_cast<const FrameDesc*>(dp) + f0;
        } else {
            DTL_HIP_CHECK(hipMemcpyAsync(d_frames_.as<FrameDesc>() + f0, h_frames_.as<FrameDesc>() + f0,
                                         size_t(f1 - f0) * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        }
        if (!payload_started && ev_) { DTL_HIP_CHECK(hipEventRecord(ev_[0], st)); payload_started = true; }
        size_t kc_a = 0, kc_m = 0, kc_b = 0;
        hipEvent_t between = nullptr;
        if (kc_.on) {
            DTL_TRY(kc_.mark(st, &kc_a));
            if (kc_.used == kc_.pool.size()) { hipEvent_t e = nullptr; DTL_HIP_CHECK(hipEventCreate(&e)); kc_.pool.push_back(e); }
            kc_m = kc_.used++;                       // recorded by launch_payload between its two kernels
            between = kc_.pool[kc_m];
        }
        DTL_TRY(launch_payload(tab_, iq, W, d_frames_.as<FrameDesc>() + f0, f1 - f0, d_segs_.as<CfoSegment>(), n_segs,
                               d_taps.as<cf32>(), d_stats_.as<FrameStat>() + f0,
                               d_llr.as<float>(), eh ? eh + size_t(f0) * L * N : nullptr, es ? es + size_t(f0) * L * N : nullptr,
                               ss_out ? ss_out + size_t(f0) * capacity_ : nullptr, (flags & RX_WANT_SOFT_SYMS) != 0,
                               nofec_bytes ? nofec_bytes + size_t(f0) * nofec_stride : nullptr,
                               nofec_inf ? nofec_inf + size_t(f0) * 2 : nullptr, nofec_stride, st, host_descs, between));
        if (kc_.on) {
            DTL_TRY(kc_.mark(st, &kc_b));
            const bool fused = between && payload_is_fused(tab_, eh || es);
            if (fused) kc_.span(KernelClock::FRAME_EQ, kc_a, kc_b);          // one kernel: the event between was never recorded
            else { kc_.span(KernelClock::FRAME_EQ, kc_a, kc_m); kc_.span(KernelClock::DEMAP, kc_m, kc_b); }
        }
        f0 = f1;
        return DTL_OK;
    };
    for (int ci = 0; ci < n_cand && !stop; ci++) {
        const int sl = slot[ci];
        // records of this candidate's launch (and of every launch before it) on the host?
        while (ready < c.hdr_launches.size() && sl >= c.hdr_launches[ready].s0) {
            DTL_TRY(flush_payload());
            DTL_HIP_CHECK(hipEventSynchronize(hdr_ev_[ready]));
            if (ready == 0) { stage_ms[2] = now_ms() - t0; t0 = now_ms(); }
            const Call::HdrLaunch& hl = c.hdr_launches[ready];
            std::memcpy(recs.data() + hl.s0, h_recs_.as<HeaderRec>() + hl.s0, size_t(hl.s1 - hl.s0) * sizeof(HeaderRec));
            ready++;
        }
        // phase B does not wait for the end of the walk (nor for a launch boundary: one early candidate with a late
        // record slot -- a trigger frame_detect_bb moved -- consumes all of them at once): the frames found so far go out
        // after the first 2048 and then every 8192
        if (int(fdesc.size()) - f0 >= (f0 == 0 ? flush_first_ : flush_frames_)) DTL_TRY(flush_payload());
        const int64_t trig = cand_abs[ci];
        n_walked = ci + 1;
        if (trig < ss.demux_pos) continue; // consumed by a previous frame
        const HeaderRec& r = recs[sl];
        FrameInfo fi{};
        fi.trigger = trig; fi.header_ok = r.crc_ok; fi.carr_offset = r.carr_offset; fi.payload_index = -1; fi.cand = sl;
        fi.llr_offset = -1;
        if (!r.crc_ok) { frames.push_back(fi); ss.demux_pos = trig + 1; continue; }
        if (trig + int64_t(hdr_syms + L) * P > E) { stop = true; break; } // only at the end of a stream
        if (r.cnst && r.cnst <= 4) ss.state_cnst = r.cnst; // :269-273
        const int bps = bits_per_symbol(ss.state_cnst);
        int n_syms = r.payload_len * 8 / bps;
        if (r.payload_len * 8 % bps) n_syms++;
        fi.payload_len = r.payload_len; fi.frame_no = r.frame_no; fi.cnst = ss.state_cnst; fi.fb_cnst = r.fb_cnst;
        fi.tb_no = r.tb_no; fi.fec_fb = r.fec_fb; fi.tb_offset = r.tb_offset; fi.fec = r.fec; fi.tb_payload = r.tb_payload;
        fi.n_payload_syms = n_syms;
        fi.payload_index = int(fdesc.size());
        FrameDesc d{};
        d.trigger = trig - A; d.seg = cands[ci].seg; d.cand = sl; d.cnst = ss.state_cnst; d.carr_offset = r.carr_offset;
        // ofdm_serializer_vcc truncation to packet_length; payload == 0 frames are dropped (:218,254)
        d.n_syms = r.payload_len ? std::min(n_syms, capacity_) : 0;
        // soft-bit stream layout: frame after frame, behind the soft bits carried for an unfinished TB
        fi.llr_offset = int64_t(llr_total);
        llr_off.push_back(uint32_t(llr_total));
        d.llr_off = uint32_t(llr_total);
        llr_total += uint64_t(d.n_syms) * bps;
        if (llr_total + 4096 > 0xFFFFFFF0ull) { set_last_error("process: soft-bit stream exceeds 32-bit offsets; split the batch"); return DTL_ERR_CAPACITY; }
        fdesc.push_back(d);
        frames.push_back(fi);
        // header_payload_demux consumes all but the LAST payload item: it may hold the next trigger
        ss.demux_pos = trig + int64_t(hdr_syms + L) * P - 1;
    }
    DTL_TRY(flush_payload());
    if (!n_cand) { stage_ms[2] = now_ms() - t0; t0 = now_ms(); }
    if (payload_started && ev_) { DTL_HIP_CHECK(hipEventRecord(ev_[1], st)); ev_mask_[(n_calls_ - 1) % kEvRing] |= 1u; }
    if (fdesc.size()) { // per-frame SNR estimates travel back while the decoder runs
        DTL_TRY(h_stats_.ensure(fdesc.size() * sizeof(FrameStat)));
        DTL_HIP_CHECK(hipMemcpyAsync(h_stats_.p, d_stats_.p, fdesc.size() * sizeof(FrameStat), hipMemcpyDeviceToHost, st));
        if (!ev_stats_) DTL_HIP_CHECK(hipEventCreateWithFlags(&ev_stats_, hipEventDisableTiming));
        DTL_HIP_CHECK(hipEventRecord(ev_stats_, st));
    }
    ss.pending_fixed.erase(ss.pending_fixed.begin(), ss.pending_fixed.begin() + n_walked);
    n_frames = int(fdesc.size());
    n_payload_frames = n_frames;
    n_llr = int64_t(llr_total);
    stats.assign(size_t(n_frames), FrameStat{});
    nofec_info.assign(size_t(n_frames) * 2, 0);
    total_bytes = 0;
    if (!n_cand && ss.llr_carry_len) { // nothing new, but a transport block is still waiting for its frames
        DTL_TRY(d_llr.ensure((size_t(llr_total) + 4096) * 4));
        DTL_HIP_CHECK(hipMemcpyAsync(d_llr.p, d_llr_carry_.p, size_t(ss.llr_carry_len) * 4, hipMemcpyDeviceToDevice, st));
    }
    return DTL_OK;
}

// transport blocks: reassembly tables, LDPC, CRC, compaction; per-frame bookkeeping
int Receiver::stage_blocks(Call& c)
{
    RX_CALL_LOCALS(c);
    TbAssembler& asmb = ss.asmb;
    asmb.done.clear();
    if (cfg.has_fec) {
        int max_n = 0, max_bps = 0;
        for (auto& d : decoders_) if (d) max_n = std::max(max_n, d->code.n);
        for (auto& m : cfg.mcs) max_bps = std::max(max_bps, m.cnst); // python/dtl/ofdm_adaptive.py:8
        asmb.rcv_cap = 2L * max_n * compute_tb_len(max_n, capacity_ * max_bps); // tb_decoder.cc:24-27
        std::vector<CodeDims> dims(decoders_.size());
        for (size_t i = 1; i < decoders_.size(); i++) if (decoders_[i]) dims[i] = { decoders_[i]->code.n, decoders_[i]->code.k };
        size_t fidx = 0;
        for (auto& fi : frames) {
            if (fi.payload_index < 0) continue;
            const FrameDesc& d = fdesc[fidx];
            const uint32_t base = llr_off[fidx];
            fidx++;
            DTL_TRY(tb_process_frame(asmb, fi, d.cnst, d.n_syms, base, capacity_, dims));
        }
    }
    // irregular blocks: their pieces go into contiguous runs behind the call's soft bits, the recipe then has one piece
    std::vector<LlrCopy> llr_copies;
    if (asmb.irregular) {
        uint64_t at = uint64_t(n_llr);
        for (auto& d : asmb.done) {
            if (d.pieces.empty()) continue;
            const uint64_t run = at;
            for (const auto& sg : d.pieces) {
                llr_copies.push_back({ sg.off, uint32_t(at), sg.len, 0 });
                at += uint64_t(sg.len);
            }
            if (at + 4096 > 0xFFFFFFF0ull) { set_last_error("process: soft-bit stream exceeds 32-bit offsets; split the batch"); return DTL_ERR_CAPACITY; }
            d.g.n_seg = 1;
            d.g.seg_off[0] = uint32_t(run); d.g.seg_len[0] = int32_t(at - run);
            for (int i = 1; i < 4; i++) { d.g.seg_off[i] = 0xFFFFFFFFu; d.g.seg_len[i] = 0; }
            d.pieces.clear();
        }
        asmb.irregular = false;
        const size_t need = (size_t(at) + 4096) * 4;
        if (need > d_llr.cap) { // rare: grow the stream buffer, keeping what the payload kernels have written
            DevBuf bigger;
            DTL_TRY(bigger.ensure(need + need / 4));
            if (d_llr.p && n_llr) DTL_HIP_CHECK(hipMemcpyAsync(bigger.p, d_llr.p, size_t(n_llr) * 4, hipMemcpyDeviceToDevice, st));
            DTL_HIP_CHECK(hipStreamSynchronize(st));
            std::swap(d_llr.p, bigger.p);
            std::swap(d_llr.cap, bigger.cap);
        }
    }
    // global codeword order: grouped per LDPC code, TB order inside a group
    n_tbs = int(asmb.done.size());
    std::vector<TbGather> gathers(n_tbs);
    std::vector<std::pair<int, std::pair<size_t, size_t>>> code_ranges; // code -> [first, count)
    uint32_t bits_total = 0;
    size_t n_cw_total = 0;
    uint64_t max_out_bytes = 0;
    int stride = 16;
    {
        uint32_t bits_base = 0;
        for (int tix = 0; tix < n_tbs; tix++) {
            asmb.done[tix].g.bits_off = bits_base;
            bits_base += (uint32_t(asmb.done[tix].g.tb_payload) + 15u) & ~7u;
            max_out_bytes += uint64_t(asmb.done[tix].g.tb_payload) / 8;
            stride = std::max<int>(stride, asmb.done[tix].g.tb_payload / 8 + 8);
        }
        for (int code = 1; code < int(decoders_.size()); code++) {
            const size_t first = n_cw_total;
            for (int tix = 0; tix < n_tbs; tix++) {
                auto& d = asmb.done[tix];
                if (d.code != code) continue;
                d.g.cw_first = uint32_t(n_cw_total);
                n_cw_total += size_t(d.g.n_cw);
            }
            if (n_cw_total > first) code_ranges.push_back({ code, { first, n_cw_total - first } });
        }
        for (int tix = 0; tix < n_tbs; tix++) {
            gathers[tix] = asmb.done[tix].g;
        }
        n_codewords = int64_t(n_cw_total);
        stride = (stride + 15) & ~15;
        DTL_TRY(d_cw_bits.ensure(size_t(bits_base) + 4096));
        bits_total = bits_base;
    }
    stage_ms[3] = now_ms() - t0; t0 = now_ms();

    // ---- LDPC -> TB check -> compaction (the payload kernel is already running)
    tb_res.assign(size_t(n_tbs), TbResult{});
    DTL_TRY(h_small_.ensure(128));
    *reinterpret_cast<uint32_t*>(h_small_.as<uint8_t>() + 96) = 0; // total bytes (filled by an async D2H below)
    DTL_TRY(d_total_.ensure(16));
    if (cfg.has_fec) {
        if (n_tbs) {
            DTL_TRY(d_cwdesc_.ensure(n_cw_total * sizeof(CwDesc)));
            DTL_TRY(d_cw_nit.ensure(n_cw_total * 4));
            DTL_TRY(d_tbres_.ensure(size_t(n_tbs) * sizeof(TbResult)));
            DTL_TRY(h_tbres_.ensure(size_t(n_tbs) * sizeof(TbResult)));
            DTL_TRY(d_len_.ensure(size_t(n_tbs) * 4));
            DTL_TRY(d_offs_.ensure(size_t(n_tbs) * 4));
            DTL_TRY(d_staging_.ensure(size_t(n_tbs) * stride));
            // What the decoder needs besides the soft bits -- the blocks' gather recipes on the device, their expansion into
            // codeword descriptors, the zeroed output words -- depends on the host's bookkeeping only, which is done while
            // phase B is still running: it goes onto the side stream and executes UNDER the demapper's last launches (on the
            // main stream these two copies, one small kernel and one fill sat between phase B and the decoder with the device
            // waiting: 0.18 ms per step).  The decoder waits for the side stream's event.
            // the decoder packs its decisions (1 bit per bit) when every code in use runs on the node-parallel kernel:
            // an eighth of the bytes out and back in; the buffer is zeroed first (the kernel ORs words in)
            bool packed = true;
            for (auto& cr : code_ranges) packed = packed && decoders_[cr.first]->packed_output_ok();
            hipStream_t pre = st;
            if (side_ && st != side_) {
                pre = side_;
                if (!ev_pre_) DTL_HIP_CHECK(hipEventCreateWithFlags(&ev_pre_, hipEventDisableTiming));
            }
            DTL_TRY(upload_vec(h_gather_, d_tbgather_, gathers, pre));
            DTL_TRY(launch_expand_cw(d_tbgather_.as<TbGather>(), n_tbs, d_cwdesc_.as<CwDesc>(), pre));
            if (packed) DTL_HIP_CHECK(hipMemsetAsync(d_cw_bits.p, 0, (size_t(bits_total) / 8 + 64 + 7) & ~size_t(7), pre));
            if (pre != st) {
                DTL_HIP_CHECK(hipEventRecord(ev_pre_, pre));
                DTL_HIP_CHECK(hipStreamWaitEvent(st, ev_pre_, 0));
            }
            if (!llr_copies.empty()) {                // (rare: soft bits moved into one run -- they come from phase B: main stream)
                DTL_TRY(upload_vec(h_llrcopy_, d_llrcopy_, llr_copies, st));
                DTL_TRY(launch_llr_gather(d_llrcopy_.as<LlrCopy>(), int(llr_copies.size()), d_llr.as<float>(), st));
            }
            if (ev_) DTL_HIP_CHECK(hipEventRecord(ev_[2], st));
            {
                // (several receivers on this device: decoders one after the other -- DecoderGate; the lock covers the launch
                // only, the dependency is the device's business)
                std::unique_lock<std::mutex> gate_lock;
                if (decoder_gate) {
                    gate_lock = std::unique_lock<std::mutex>(decoder_gate->mu);
                    if (decoder_gate->last_done) DTL_HIP_CHECK(hipStreamWaitEvent(st, decoder_gate->last_done, 0));
                    if (ev_) DTL_HIP_CHECK(hipEventRecord(ev_[2], st));   // (the decoder's start, behind the wait)
                }
                for (auto& cr : code_ranges) {
                    KC_BEGIN(st);
                    DTL_TRY(decoders_[cr.first]->decode(d_llr.as<float>(), d_cwdesc_.as<CwDesc>() + cr.second.first, cr.second.second,
                                                        d_cw_bits.as<uint8_t>(), d_cw_nit.as<int32_t>() + cr.second.first, st, packed));
                    KC_END(KernelClock::LDPC, st);
                }
                if (ev_) { DTL_HIP_CHECK(hipEventRecord(ev_[3], st)); ev_mask_[(n_calls_ - 1) % kEvRing] |= 2u; }
                if (decoder_gate) {
                    hipEvent_t& mine = ev_dec_[dec_launches_++ & 1u];
                    if (!mine) DTL_HIP_CHECK(hipEventCreateWithFlags(&mine, hipEventDisableTiming));
                    DTL_HIP_CHECK(hipEventRecord(mine, st));
                    decoder_gate->last_done = mine;
                }
            }
            DTL_TRY(d_payload.ensure(max_out_bytes + 64));
            KC_BEGIN(st);
            DTL_TRY(launch_tb_check(d_tbgather_.as<TbGather>(), n_tbs, d_cw_bits.as<uint8_t>(), d_cw_nit.as<int32_t>(),
                                    d_staging_.as<uint8_t>(), stride, d_tbres_.as<TbResult>(), d_len_.as<int32_t>(), st, packed));
            DTL_TRY(launch_compact(d_staging_.as<uint8_t>(), stride, d_len_.as<int32_t>(), n_tbs, d_offs_.as<uint32_t>(),
                                   d_total_.as<uint32_t>(), d_payload.as<uint8_t>(), st));
            KC_END(KernelClock::TB_CHECK, st);
            DTL_HIP_CHECK(hipMemcpyAsync(h_tbres_.p, d_tbres_.p, size_t(n_tbs) * sizeof(TbResult), hipMemcpyDeviceToHost, st));
            DTL_HIP_CHECK(hipMemcpyAsync(h_small_.as<uint8_t>() + 96, d_total_.p, 4, hipMemcpyDeviceToHost, st));
        }
    } else if (n_frames) {
        DTL_TRY(d_len_.ensure(size_t(n_frames) * 4));
        DTL_TRY(d_offs_.ensure(size_t(n_frames) * 4));
        DTL_TRY(d_payload.ensure(size_t(n_frames) * nofec_stride + 64));
        DTL_TRY(launch_nofec_lengths(d_nofec_info_.as<int32_t>(), n_frames, d_len_.as<int32_t>(), st));
        DTL_TRY(launch_compact(d_staging_.as<uint8_t>(), nofec_stride, d_len_.as<int32_t>(), n_frames, d_offs_.as<uint32_t>(),
                               d_total_.as<uint32_t>(), d_payload.as<uint8_t>(), st));
        DTL_TRY(h_nofec_.ensure(size_t(n_frames) * 8));
        DTL_HIP_CHECK(hipMemcpyAsync(h_nofec_.p, d_nofec_info_.p, size_t(n_frames) * 8, hipMemcpyDeviceToHost, st));
        DTL_HIP_CHECK(hipMemcpyAsync(h_small_.as<uint8_t>() + 96, d_total_.p, 4, hipMemcpyDeviceToHost, st));
    }
    // ---- per-frame sequential bookkeeping while the decoder is still running: lost frames (:124-138),
    // feedback decision
    if (n_frames) {
        DTL_HIP_CHECK(hipEventSynchronize(ev_stats_));
        std::memcpy(stats.data(), h_stats_.p, size_t(n_frames) * sizeof(FrameStat));
    }
    {
        if (!ss.fb_init) { ss.fb.init(cfg.mcs, cfg.hysteresis, cfg.decision_th, cfg.initial_mcs); ss.fb_init = true; }
        size_t fidx = 0;
        for (auto& fi : frames) {
            if (fi.payload_index < 0) continue;
            const FrameStat& s = stats[fidx++];
            int lost = 0;
            if (fi.frame_no != ss.expected_frame_no)
                lost = fi.frame_no < ss.expected_frame_no ? 4096 + fi.frame_no - ss.expected_frame_no : fi.frame_no - ss.expected_frame_no;
            ss.lost_total += lost;
            ss.expected_frame_no = (fi.frame_no + 1) % 4096;
            fi.snr_db = s.snr_db; fi.noise = s.noise_lin;
            ss.fb.get(s.snr_db, &fi.feedback_cnst, &fi.feedback_fec);
            fi.lost_frames = int(ss.lost_total);
        }
    }
    // the block records' static half (number, announced length) while the device is still busy; stage_records adds
    // the verdicts
    if (cfg.has_fec) {
        tbs.resize(size_t(n_tbs));
        for (int tix = 0; tix < n_tbs; tix++) {
            TbInfo ti{};
            ti.tb_no = asmb.done[tix].tb_no; ti.tb_payload = asmb.done[tix].tb_payload;
            tbs[size_t(tix)] = ti;
        }
    }
    DTL_HIP_CHECK(hipStreamSynchronize(st));
    if (kc_.on) DTL_TRY(kc_.collect());
    if (cfg.has_fec && n_tbs) std::memcpy(tb_res.data(), h_tbres_.p, size_t(n_tbs) * sizeof(TbResult));
    if (!cfg.has_fec && n_frames) std::memcpy(nofec_info.data(), h_nofec_.p, size_t(n_frames) * 8);
    total_bytes = *reinterpret_cast<uint32_t*>(h_small_.as<uint8_t>() + 96);
    // the kernels' device times: on request only (two event reads per pair; callers that time several steps read them
    // afterwards through kernel_times)
    stage_ms[6] = stage_ms[7] = 0.0;
    if (flags & RX_WANT_KERNEL_MS) {
        double l = -1, p2 = -1;
        (void)kernel_times(0, &l, &p2);
        if (l >= 0) stage_ms[6] = l;
        if (p2 >= 0) stage_ms[7] = p2;
    }
    n_payload_bytes = total_bytes;
    stage_ms[4] = now_ms() - t0; t0 = now_ms();

    return DTL_OK;
}

// per-block records of the result
int Receiver::stage_records(Call& c)
{
    RX_CALL_LOCALS(c);
    // ---- TB records
    if (cfg.has_fec) {
        int64_t out_off = 0;
        for (int tix = 0; tix < n_tbs; tix++) {
            TbInfo& ti = tbs[size_t(tix)];                  // number and announced length: filled in stage_blocks
            ti.crc_ok = tb_res[tix].crc_ok; ti.avg_it = tb_res[tix].avg_it; ti.user_bits = tb_res[tix].user_bits;
            ti.out_offset = ti.crc_ok ? out_off : -1;
            if (ti.crc_ok) out_off += ti.user_bits / 8;
        }
    } else {
        int64_t out_off = 0;
        size_t fidx = 0;
        for (auto& fi : frames) {
            if (fi.payload_index < 0) continue;
            const size_t f = fidx++;
            if (fi.payload_len == 0) continue;
            TbInfo ti{};
            const int nw = nofec_info[2 * f + 1];
            ti.tb_no = fi.frame_no; ti.tb_payload = nw; ti.crc_ok = nofec_info[2 * f]; ti.avg_it = 0;
            ti.user_bits = (nw - 4) * 8; ti.out_offset = out_off;
            if (nw >= 4) out_off += nw - 4;
            tbs.push_back(ti);
        }
    }
    stage_ms[5] = now_ms() - t0;
    if (dbg) fprintf(stderr, "  [n_edges=%u plateau %.3f ms]\n", dbg_n_edges, dbg_plateau_ms);
    return DTL_OK;
}

// what the next call needs from this one
int Receiver::stream_end(Call& c)
{
    RX_CALL_LOCALS(c);
    TbAssembler& asmb = ss.asmb;

    // ---- what the next call needs from this one
    // (a) soft bits of the transport block still being received, compacted into d_llr_carry_
    if (cfg.has_fec && !asmb.rcv.empty()) {
        size_t tot = 0;
        for (const auto& sg : asmb.rcv) if (sg.off != 0xFFFFFFFFu) tot += size_t(sg.len);
        DTL_TRY(d_llr_carry_.ensure(std::max<size_t>(tot, 1) * 4));
        size_t at = 0;
        std::vector<TbAssembler::Seg> merged; // pieces that end up adjacent in the carry become one segment
        for (const auto& sg : asmb.rcv) {
            if (sg.off == 0xFFFFFFFFu) { merged.push_back(sg); continue; }
            DTL_HIP_CHECK(hipMemcpyAsync(d_llr_carry_.as<float>() + at, d_llr.as<float>() + sg.off, size_t(sg.len) * 4,
                                         hipMemcpyDeviceToDevice, st));
            if (!merged.empty() && merged.back().off != 0xFFFFFFFFu) merged.back().len += sg.len;
            else merged.push_back({ uint32_t(at), sg.len });
            at += size_t(sg.len);
        }
        asmb.rcv.swap(merged);
        ss.llr_carry_len = int64_t(tot);
    } else {
        ss.llr_carry_len = 0;
    }
    // (b) samples: from the first trigger whose frame is still incomplete (the demux delays the stream
    // by one symbol), and in any case the history the timing metric / arg(P) windows of the next
    // samples reach back to; rounded down to the sync kernel's tile
    {
        auto& pf = ss.pending_fixed;
        pf.erase(std::remove_if(pf.begin(), pf.end(), [&](int64_t t) { return t < ss.demux_pos; }), pf.end());
        const int64_t hist = 2L * N + cp + 2;
        int64_t keep = produced - hist;
        if (!pf.empty()) keep = std::min(keep, pf.front() - P);
        // a plateau that never ends (e.g. a constant input) must not grow the carry without bound
        const int64_t max_carry = 4L * frame_samples() + hist;
        if (E - keep > max_carry) {
            keep = E - max_carry;
            pf.erase(std::remove_if(pf.begin(), pf.end(), [&](int64_t t) { return t - P < keep; }), pf.end());
        }
        // the tile of the sync kernel that contains the next new sample must see real history, not the
        // zero fill of a buffer start: then every metric value equals the one a single batch computes
        keep = std::min(keep, (E & ~int64_t(kSyncTileSamples - 1)) - N - N / 2);
        keep = std::max<int64_t>(std::max<int64_t>(keep, A), 0);
        keep &= ~int64_t(kSyncTileSamples - 1);
        const int64_t len = E - keep;
        DTL_TRY(d_carry_.ensure(std::max<size_t>(size_t(len), 1) * sizeof(cf32)));
        if (len) DTL_HIP_CHECK(hipMemcpyAsync(d_carry_.p, iq + (keep - A), size_t(len) * sizeof(cf32), hipMemcpyDeviceToDevice, st));
        ss.carry_abs = keep;
        ss.carry_len = len;
        // oscillator segments: the one active at `keep - P` and everything after it
        size_t first = 0;
        while (first + 1 < ss.segs.size() && ss.segs[first + 1].start <= keep - P) first++;
        if (first) ss.segs.erase(ss.segs.begin(), ss.segs.begin() + first);
    }
    DTL_HIP_CHECK(hipStreamSynchronize(st)); // the caller may reuse its buffer; d_work_ is reused by the next call
    ss.next_abs = E;
    return DTL_OK;
}

int Receiver::process_stream(const cf32* d_iq, int64_t n, unsigned flags, bool final, hipStream_t st)
{
    DeviceScope dev;                     // dtl_rx_cfg.device: current for this call, the caller's own afterwards
    DTL_TRY(dev.enter(cfg.device));
    return process_stream_body(d_iq, n, flags, final, st);
}

int Receiver::process_stream_body(const cf32* d_iq, int64_t n, unsigned flags, bool final, hipStream_t st)
{
    StreamState& ss = *ss_;
    struct ResetOnExit { // a failed or final call ends the stream
        Receiver* r; bool armed;
        ~ResetOnExit() { if (armed) r->reset_stream(); }
    } reset_guard{ this, true };
    if (n < 0 || (!d_iq && n > 0)) { set_last_error("process: bad input"); return DTL_ERR_INVALID_ARG; }
    if (cfg.has_fec && decoders_.size() < 2) { set_last_error("Misconfiguration: decoder not found!"); return DTL_ERR_INVALID_ARG; }
    {   // the header fields (32 bits, + 48 with FEC, + CRC-16) must fit the header symbols' data carriers: the reference's
        // formatter would silently cut them off (packet_header.cc put_field guards) and nothing decodes
        int bits = 0;
        for (int i = 0; i < cfg.header_len(); i++) bits += int(cfg.occupied_carriers[size_t(i) % cfg.occupied_carriers.size()].size());
        if (bits < (cfg.has_fec ? 96 : 48)) { set_last_error("header symbols hold fewer bits than the header fields"); return DTL_ERR_INVALID_ARG; }
    }
    frames.clear(); tbs.clear(); raw_triggers.clear(); fixed_triggers.clear(); trigger_angles.clear();
    n_payload_frames = n_payload_bytes = n_llr = n_codewords = 0;
    for (double& s : stage_ms) s = 0.0;
    const int64_t W = ss.carry_len + n; // samples this call works on: carried tail + new ones
    if (W == 0 || (n == 0 && !final)) { reset_guard.armed = final; return DTL_OK; } // nothing consumed, nothing produced
    if (W > int64_t(0x7fffffff)) { set_last_error("process: at most 2^31-1 samples per call"); return DTL_ERR_INVALID_ARG; }
    if (!uploaded_) DTL_TRY(upload_tables());
    {
        const int slot = int(n_calls_++ % kEvRing);
        if (!ev_ring_[slot][0]) for (auto& e : ev_ring_[slot]) DTL_HIP_CHECK(hipEventCreate(&e));
        ev_ = ev_ring_[slot];
        ev_mask_[slot] = 0;
    }
    kc_.begin((flags & RX_WANT_KERNEL_SPLIT) != 0);
    auto own = std::make_unique<Call>();
    Call& c = *own;
    c.d_iq = d_iq; c.iq = d_iq; c.n = n; c.W = W; c.flags = flags; c.final = final; c.st = st;
    // A = absolute index of the working buffer's first sample (a multiple of the sync kernel's tile, so that
    // every metric value is computed exactly as in one batch); new samples are [B, E)
    c.A = ss.carry_len ? ss.carry_abs : ss.next_abs;
    c.B = ss.next_abs; c.E = c.B + n;
    c.N = cfg.fft_len; c.cp = cfg.cp_len; c.P = c.N + c.cp; c.h = cfg.header_len(); c.L = cfg.frame_length;
    c.hdr_syms = 2 + c.h;
    c.dbg = getenv("DTL_DEBUG") != nullptr;
    c.tdbg = c.t0 = now_ms();
    DTL_TRY(stream_begin(c));
    DTL_TRY(stage_sync(c));
    DTL_TRY(stage_triggers(c));
    DTL_TRY(stage_frames(c));
    DTL_TRY(stage_blocks(c));
    DTL_TRY(stage_records(c));
    if (final) return DTL_OK; // reset_guard ends the stream
    DTL_TRY(stream_end(c));
    reset_guard.armed = false;
    return DTL_OK;
}

} // namespace dtl
