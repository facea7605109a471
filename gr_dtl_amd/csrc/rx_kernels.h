// rx_kernels.h -- launchers of the receive-path kernels (rx_kernels.hip).
#pragma once

#include "dtl_common.h"
#include "ldpc_bp.h"
#include "rx_types.h"

namespace dtl {

// outputs per workgroup of the sync kernel; a continued stream re-joins its carried samples at a multiple
// of this so that every timing-metric value is recomputed bit-identically
constexpr int kSyncTileSamples = 4096;

// edges / metric are produced for samples m_begin .. n-1 only (earlier samples are history of a stream
// continued from a previous call); metric_out[m - m_begin]
// check (tests / tools): three counters -- strips, strips the single-precision pre-test left to the exact path, outputs
// the pre-test decided differently from the exact test (must be 0); every strip then runs both
int launch_sync_edges(const cf32* iq, int64_t n, int64_t m_begin, int fft_len, float threshold,
                      unsigned long long* edges, unsigned int* n_edges, unsigned int max_edges, float* metric_out,
                      hipStream_t st, unsigned long long* check = nullptr);
int launch_trigger_angle(const cf32* iq, int64_t n, int fft_len, const int64_t* trig, int n_trig,
                         float* angle, hipStream_t st);
int launch_header(const RxTables& t, const cf32* iq, int64_t n, const Candidate* cands, int n_cand,
                  const CfoSegment* segs, int n_segs, HeaderRec* recs, cf32* taps, uint8_t* hdr_bits,
                  hipStream_t st);
int launch_payload(const RxTables& t, const cf32* iq, int64_t n, const FrameDesc* frames, int n_frames,
                   const CfoSegment* segs, int n_segs, const cf32* taps, FrameStat* stats,
                   float* llr, cf32* eq_hard, cf32* eq_soft, cf32* soft_syms, bool soft_syms_wanted,
                   uint8_t* nofec_bytes, int32_t* nofec_info, int nofec_stride, hipStream_t st, const FrameDesc* frames_host = nullptr,
                   hipEvent_t between = nullptr /* recorded between the frame equaliser and the kernel behind it */);
// phase B as one kernel (soft demapper in the frame equaliser's epilogue: no soft_syms hand-over buffer needed)?
bool payload_is_fused(const RxTables& t, bool eq_rows);
int launch_tb_check(const TbGather* tbs, int n_tbs, const uint8_t* bits, const int32_t* nit, uint8_t* staging,
                    int stride, TbResult* res, int32_t* len, hipStream_t st, bool packed = false);
int launch_expand_cw(const TbGather* tbs, int n_tbs, CwDesc* out, hipStream_t st);
int launch_llr_gather(const LlrCopy* ops, int n_ops, float* llr, hipStream_t st);
int launch_nofec_lengths(const int32_t* info, int n, int32_t* len, hipStream_t st);
int launch_compact(const uint8_t* staging, int stride, const int32_t* len, int n, uint32_t* offs,
                   uint32_t* total, uint8_t* out, hipStream_t st);
// per-block entry points on frequency-domain vectors (dtl_chanest, dtl_header_decode, dtl_equalize_frames)
int launch_header_stage(const RxTables& t, int n, const HeaderStageIo& io, HeaderRec* recs, cf32* taps_out,
                        uint8_t* hdr_bits, hipStream_t st);
int launch_frame_eq_stage(const RxTables& t, const cf32* freq, const FrameDesc* frames, int n_frames, const cf32* taps,
                          FrameStat* stats, cf32* eq_hard, cf32* eq_soft, cf32* soft_syms, hipStream_t st);
// *d_bad += something non-zero when a float of x[0..n) is outside {NaN, +-0, normal with |x| <= 2^62}
int launch_range_check(const float* x, int64_t n, unsigned int* d_bad, hipStream_t st);
// stand-alone stages (block shims, unit tests)
void constellation_points_host(int cnst, cf32* out16);
int launch_soft_demap(const cf32* syms, int64_t n, int cnst, double noise, int dist_mode, float* llr, hipStream_t st);
int launch_fft_batch(const cf32* in, cf32* out, int64_t n_vec, int fft_len, int forward, hipStream_t st);
int launch_crc32(const uint8_t* bytes, int64_t len, uint32_t* d_out, hipStream_t st);
size_t header_lds_bytes(const RxTables& t);
size_t payload_lds_bytes(const RxTables& t);

} // namespace dtl
