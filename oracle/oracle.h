/*
 * oracle.h -- CPU restatement ("oracle") of gr-dtl's adaptive-OFDM receive path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link,
 * import or call it -- and there only as the checker / reported CPU baseline.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * gr-dtl tree).  The half of the path that lives in GNU Radio 3.10 (not present in
 * the reference tree, not installable here) is restated from GR 3.10 semantics; those
 * functions are marked "GR3.10 (unverified restatement)".
 *
 * PARITY PINNING: the oracle is pinned against every known answer the reference's own
 * tests hold for this path (tests/golden/ JSON files, see tests/test_oracle_golden.py):
 * header bit vectors + CRC-16, LSB-first repack vector, 8PSK/BPSK metric constant,
 * feedback-decision sequence, QPSK map/slice identity and the TX->RX round trips.
 * Values that no reference test fixes (LLR magnitudes, BP iterations, SNR estimates,
 * Schmidl-Cox metric, channel taps, cldpc permutation) are "parity unpinned".
 */
#ifndef DTL_ORACLE_H
#define DTL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_c32;

/* ------------------------------------------------------------------ bits / crc */

/* lib/testbed/repack.cc:31-69 (stateless form: indexes start at 0, as a fresh object) */
int orc_repack_lsb_first(const uint8_t* in, size_t n_in, uint8_t* out,
                         int bits_in, int bits_out, int full_output_symbols);
/* lib/testbed/repack.cc:72-112 */
int orc_repack_msb_first(const uint8_t* in, size_t n_in, uint8_t* out,
                         int bits_in, int bits_out, int full_output_symbols);

/* gr::digital::crc (GR3.10) generic bitwise CRC, as instantiated at
 * lib/dtl/crc_util.cc:23 (32,0x04C11DB7,0xFFFFFFFF,0xFFFFFFFF,refin,refout) and
 * lib/dtl/ofdm_adaptive_packet_header.cc:72 (16,0x1021,0xFFFF,0,false,true). */
uint64_t orc_crc_compute(int num_bits, uint64_t poly, uint64_t init, uint64_t final_xor,
                         int input_reflected, int result_reflected,
                         const uint8_t* data, size_t len);
/* lib/dtl/crc_util.cc:29-36 : CRC-32 appended little-endian */
uint32_t orc_crc32_append(uint8_t* buf, size_t len);
/* lib/dtl/crc_util.cc:41-61 */
int orc_crc32_verify(const uint8_t* buf, size_t len_with_crc);

/* ------------------------------------------------------------------ constellations */

enum { ORC_CNST_UNKNOWN = 0, ORC_CNST_BPSK = 1, ORC_CNST_QPSK = 2, ORC_CNST_PSK8 = 3,
       ORC_CNST_QAM16 = 4 };

/* include/gnuradio/dtl/ofdm_adaptive_utils.h:22-28, lib/dtl/ofdm_adaptive_utils.cc:64-70 */
int orc_bits_per_symbol(int cnst);
/* points: GR3.10 constellation_{bpsk,qpsk,8psk,16qam} + lib/dtl/constellation.cc:18-24 */
int orc_constellation_points(int cnst, orc_c32* pts /* >=16 */);
/* GR3.10 decision_maker of each constellation (used at lib/dtl/ofdm_adaptive_equalizer.cc:253) */
unsigned orc_decision_maker(int cnst, orc_c32 sample);
/* GR3.10 constellation::calc_soft_dec (used at
 * lib/dtl/ofdm_adaptive_constellation_soft_cf_impl.cc:144); out[0..bps) MSB first.
 * dist_mode 0: |x-s|^2 / npwr (SURVEY 8c choice), 1: |x-s| / npwr. */
void orc_calc_soft_dec(int cnst, orc_c32 sample, float npwr, int dist_mode, float* out);
/* lib/dtl/ofdm_adaptive_constellation_soft_cf_impl.cc:143-148 (+ sigma rule :89-103):
 * n_syms symbols -> n_syms*bps LLRs, LSB first per symbol.  noise<0 => no noise tag. */
void orc_soft_demap(int cnst, const orc_c32* syms, int n_syms, double noise, int dist_mode,
                    float* llr_out);

/* ------------------------------------------------------------------ LDPC */

typedef struct orc_ldpc orc_ldpc;

/* GR3.10 alist::read + cldpc::set_alist (GF2Mat::get_G) as driven by
 * lib/dtl/ldpc_dec.cc:34-59 / lib/dtl/ldpc_enc.cc:35-56 */
orc_ldpc* orc_ldpc_load(const char* alist_path);
void orc_ldpc_free(orc_ldpc* c);
int orc_ldpc_n(const orc_ldpc* c);
int orc_ldpc_k(const orc_ldpc* c);
int orc_ldpc_m(const orc_ldpc* c);
int orc_ldpc_nedges(const orc_ldpc* c);
void orc_ldpc_permute(const orc_ldpc* c, int* out_n);
/* parity part of G after reduction: rank x k bytes, row-major (for the product's tests) */
void orc_ldpc_parity_matrix(const orc_ldpc* c, uint8_t* out_rank_by_k);
/* lib/dtl/ldpc_enc.cc:58-68 -> GR3.10 cldpc::encode; in: k bits (1/byte), out: n bits */
void orc_ldpc_encode(const orc_ldpc* c, const uint8_t* data_k, uint8_t* out_n);
/* lib/dtl/ldpc_dec.cc:62-71 -> GR3.10 awgn_bp::decode.  llr: n floats (positive = bit 1,
 * on-air order); out: k bits (1/byte).  sigma: awgn_bp's sigma (sqrt(2)
 * makes rx_lr = exp(llr)).  Returns k. */
int orc_ldpc_decode(const orc_ldpc* c, const float* llr_n, int max_it, float sigma,
                    uint8_t* out_k, int* nit);
/* batch helper (same function, ncw codewords back to back); n_threads>1 uses pthreads */
void orc_ldpc_decode_batch(const orc_ldpc* c, const float* llr, int ncw, int max_it,
                           float sigma, uint8_t* out_bits, int* nit, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
