// sharding.cpp -- frame-batch sharding across the GPUs of one node, host side of the C ABI (dtl_shard_*, dtl_owned_*).
//
// Frames are independent given their own preamble and header: the reference resets the equaliser's taps from the
// "ofdm_sync_chan_taps" tag and its SNR estimator for every frame (lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:
// 117-118,180); what crosses frames is metadata (a transport block continues in the next frame, counters).  So a stream
// is cut into contiguous runs of frames, one per GPU, each decoded by an ordinary receiver handle
// (dtl_rx_process / dtl_rx_process_stream) with a few frames of context on either side; these functions say where to
// cut, and which of the transport blocks a rank decoded are its own.  Pure host arithmetic: no device, no
// communication library -- the exchange itself belongs to the host program (RCCL point-to-point in
// gr_dtl_amd/sharding.py; a C++ host would use rccl.h the same way).
#include "../../include/dtl_rx.h"

#include "dtl_common.h"
#include "dtl_guard.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int kWarmupFrames = 4; // frame_detect_bb is in sync after three consistent triggers (kSyncedTh, ..._frame_detect_bb_impl.cc:21)

// block numbers as a non-decreasing sequence (the header field wraps; consecutive records are less than one wrap apart)
std::vector<int64_t> unwrap(const std::vector<int64_t>& numbers, int64_t modulus)
{
    std::vector<int64_t> out(numbers.size(), 0);
    for (size_t i = 1; i < numbers.size(); i++) {
        int64_t d = (numbers[i] - numbers[i - 1]) % modulus;
        if (d < 0) d += modulus;
        out[i] = out[i - 1] + d;
    }
    return out;
}

} // namespace

extern "C" {

int dtl_shard_plan(int64_t n_frames, int64_t frame_samples, int64_t lead, int64_t total_samples, int32_t world_size,
                   int32_t rank, int32_t guard_frames, dtl_shard* out)
{
    DTL_API_BEGIN("dtl_shard_plan")
    if (!out || world_size < 1 || rank < 0 || rank >= world_size || n_frames < 0 || frame_samples <= 0 || lead < 0 ||
        total_samples < 0 || guard_frames < 0) {
        dtl::set_last_error("shard_plan: bad argument");
        return DTL_ERR_INVALID_ARG;
    }
    // frame f -> rank floor(f * G / F): contiguous runs
    const int64_t lo = (n_frames * rank) / world_size, hi = (n_frames * (rank + 1)) / world_size;
    const int64_t own_lo = rank ? lead + lo * frame_samples : 0;
    const int64_t own_hi = rank + 1 < world_size ? lead + hi * frame_samples : total_samples;
    // WARMUP frames in front so that the trigger-fixing state machine is in sync when the rank's own frames begin (only
    // then does it re-insert a missed preamble the way one receiver does); guard frames behind: a transport block that
    // starts in the rank's last frame must be complete (a block slightly larger than a frame touches three frames, and
    // frames are cut at their trigger, a symbol behind the nominal frame start)
    const int64_t front = std::max<int64_t>(guard_frames, kWarmupFrames) * frame_samples;
    out->start = rank ? std::max<int64_t>(0, own_lo - front) : 0;
    out->stop = rank + 1 < world_size ? std::min<int64_t>(total_samples, own_hi + int64_t(guard_frames) * frame_samples) : total_samples;
    out->own_lo = std::min(own_lo, total_samples);
    out->own_hi = std::min(own_hi, total_samples);
    return DTL_OK;
    DTL_API_END
}

int dtl_shard_pieces(const dtl_shard* plan, int32_t n_pieces, int64_t align, int64_t* bounds, int32_t* n_out)
{
    DTL_API_BEGIN("dtl_shard_pieces")
    if (!plan || !bounds || !n_out || n_pieces < 1 || align < 1 || plan->stop < plan->start) {
        dtl::set_last_error("shard_pieces: bad argument");
        return DTL_ERR_INVALID_ARG;
    }
    // cuts at multiples of `align` samples behind the shard's first sample (the receiver's sync kernel works in tiles
    // counted from the stream's sample 0, i.e. from the shard's start); duplicates collapse: a short shard has fewer pieces
    const int64_t len = plan->stop - plan->start;
    int32_t n = 0;
    bounds[n++] = plan->start;
    for (int32_t j = 1; j < n_pieces; j++) {
        const int64_t cut = plan->start + (len * j / n_pieces) / align * align;
        if (cut > bounds[n - 1] && cut < plan->stop) bounds[n++] = cut;
    }
    bounds[n] = plan->stop;
    *n_out = len > 0 ? n : 0;
    if (len == 0) bounds[1] = plan->stop;
    return DTL_OK;
    DTL_API_END
}

int dtl_owned_blocks(const dtl_rx_frame_info* frames, int64_t n_frames, const dtl_rx_tb_info* tbs, int64_t n_tbs,
                     const dtl_shard* plan, int64_t* idx_out, int64_t cap, int64_t* n_out)
{
    DTL_API_BEGIN("dtl_owned_blocks")
    if (!plan || !n_out || n_frames < 0 || n_tbs < 0 || (n_frames && !frames) || (n_tbs && !tbs) || cap < 0 || (cap && !idx_out)) {
        dtl::set_last_error("owned_blocks: bad argument");
        return DTL_ERR_INVALID_ARG;
    }
    *n_out = 0;
    if (n_tbs == 0) return DTL_OK;
    // A frame's header names the block that STARTS in it (tb_no; tb_offset is the tail of the previous block) and blocks
    // complete in stream order: both lists become running block counts (the 8-bit number unwrapped) and are matched by value
    std::vector<int64_t> s_no, s_pos;
    for (int64_t f = 0; f < n_frames; f++) {
        const dtl_rx_frame_info& fi = frames[f];
        if (fi.payload_index < 0 || fi.payload_len <= 0) continue;
        if (!s_no.empty() && s_no.back() == fi.tb_no) continue;       // a frame that continues a block starts none
        s_no.push_back(fi.tb_no);
        s_pos.push_back(fi.trigger + plan->start);
    }
    if (s_no.empty()) return DTL_OK;
    const std::vector<int64_t> s_seq = unwrap(s_no, 256);
    std::vector<int64_t> t_no(static_cast<size_t>(n_tbs));
    for (int64_t t = 0; t < n_tbs; t++) t_no[size_t(t)] = tbs[t].tb_no;
    std::vector<int64_t> t_seq = unwrap(t_no, 256);
    int64_t first = (t_no[0] - s_no[0]) % 256;                        // the first record's block started <= 255 starts in
    if (first < 0) first += 256;
    int64_t n = 0, last_pos = -1;
    bool have_last = false;
    for (int64_t t = 0; t < n_tbs; t++) {
        const int64_t seq = t_seq[size_t(t)] + first;
        const auto it = std::lower_bound(s_seq.begin(), s_seq.end(), seq);
        int64_t pos;
        if (it != s_seq.end() && *it == seq) { pos = s_pos[size_t(it - s_seq.begin())]; last_pos = pos; have_last = true; }
        else pos = have_last ? last_pos : -1;                         // a record without a start of its own (a block flushed
                                                                      // twice around a lost frame) goes where its predecessor went
        if (pos >= plan->own_lo && pos < plan->own_hi) {
            if (n < cap) idx_out[n] = t;
            n++;
        }
    }
    *n_out = n;
    if (n > cap) { dtl::set_last_error("owned_blocks: index capacity too small"); return DTL_ERR_CAPACITY; }
    return DTL_OK;
    DTL_API_END
}

int dtl_owned_payload_ranges(const dtl_rx_tb_info* tbs, int64_t n_tbs, const int64_t* idx, int64_t n_idx, int64_t* offs,
                             int64_t* lens, int64_t cap, int64_t* n_out)
{
    DTL_API_BEGIN("dtl_owned_payload_ranges")
    if (!n_out || n_tbs < 0 || n_idx < 0 || (n_idx && (!tbs || !idx)) || cap < 0 || (cap && (!offs || !lens))) {
        dtl::set_last_error("owned_payload_ranges: bad argument");
        return DTL_ERR_INVALID_ARG;
    }
    // the CRC-ok blocks among `idx`, adjacent ones merged: the usual result is ONE range
    int64_t n = 0;
    for (int64_t i = 0; i < n_idx; i++) {
        if (idx[i] < 0 || idx[i] >= n_tbs) { dtl::set_last_error("owned_payload_ranges: index out of range"); return DTL_ERR_INVALID_ARG; }
        const dtl_rx_tb_info& t = tbs[idx[i]];
        if (t.crc_ok != 1 || t.out_offset < 0 || t.user_bits < 8) continue;
        const int64_t off = t.out_offset, len = t.user_bits / 8;
        if (n > 0 && n <= cap && offs[n - 1] + lens[n - 1] == off) { lens[n - 1] += len; continue; }
        if (n < cap) { offs[n] = off; lens[n] = len; }
        n++;
    }
    *n_out = n;
    if (n > cap) { dtl::set_last_error("owned_payload_ranges: capacity too small"); return DTL_ERR_CAPACITY; }
    return DTL_OK;
    DTL_API_END
}

} // extern "C"
