/*
 * orc_constellation.c -- ORACLE (test infrastructure, not product): constellation tables,
 * slicers and the exact-MAP soft demapper.
 *
 * gr-dtl side: lib/dtl/ofdm_adaptive_utils.cc:52-70 (registry, bits/symbol),
 * lib/dtl/constellation.cc:18-24 (QPSK scaled by 0.5),
 * lib/dtl/ofdm_adaptive_constellation_soft_cf_impl.cc:89-148 (sigma rule, bit reversal).
 * GNU Radio side (GR3.10, unverified restatement): constellation_{bpsk,qpsk,8psk,16qam}
 * tables + decision_maker, constellation::calc_soft_dec.  Pinned by the reference tests:
 * 8PSK[0] = e^{j pi/8}, BPSK[0] = -1 (qa_ofdm_adaptive_constellation_metric_vcvf.py:86),
 * QPSK map->slice identity (qa_ofdm_adaptive_chunks_to_symbols_bc.py:40,63).
 */
#include "oracle.h"

#include <math.h>

#include "orc_math.h"

int orc_bits_per_symbol(int cnst)
{
    switch (cnst) {
    case ORC_CNST_BPSK: return 1;
    case ORC_CNST_QPSK: return 2;
    case ORC_CNST_PSK8: return 3;
    case ORC_CNST_QAM16: return 4;
    default: return 0;
    }
}

int orc_constellation_points(int cnst, orc_c32* p)
{
    switch (cnst) {
    case ORC_CNST_BPSK:
        p[0] = (orc_c32){ -1, 0 };
        p[1] = (orc_c32){ 1, 0 };
        return 2;
    case ORC_CNST_QPSK: {
        /* GR: #define SQRT_TWO 0.707107 ; dtl scales by d_factor = 0.5 (double) */
        const float s = (float)((double)(float)0.707107 * 0.5);
        p[0] = (orc_c32){ -s, -s };
        p[1] = (orc_c32){ s, -s };
        p[2] = (orc_c32){ -s, s };
        p[3] = (orc_c32){ s, s };
        return 4;
    }
    case ORC_CNST_PSK8: {
        const float angle = (float)(M_PI / 8.0);
        static const int mult[8] = { 1, 7, 15, 9, 3, 5, 13, 11 };
        for (int i = 0; i < 8; i++)
            p[i] = (orc_c32){ (float)cos(mult[i] * angle), (float)sin(mult[i] * angle) };
        return 8;
    }
    case ORC_CNST_QAM16: {
        const float level = sqrtf((float)0.1);
        /* index bits: b0 = outer imag, b1 = outer real, b2 = real<0, b3 = imag>0 */
        for (int i = 0; i < 16; i++) {
            float re = ((i & 2) ? 3 : 1) * level * ((i & 4) ? -1.f : 1.f);
            float im = ((i & 1) ? 3 : 1) * level * ((i & 8) ? 1.f : -1.f);
            p[i] = (orc_c32){ re, im };
        }
        return 16;
    }
    default: return 0;
    }
}

unsigned orc_decision_maker(int cnst, orc_c32 s)
{
    switch (cnst) {
    case ORC_CNST_BPSK: return s.re > 0;
    case ORC_CNST_QPSK: return 2 * (s.im > 0) + (s.re > 0);
    case ORC_CNST_PSK8: {
        unsigned ret = 0;
        if (fabsf(s.re) <= fabsf(s.im)) ret = 4;
        if (s.re <= 0) ret |= 1;
        if (s.im <= 0) ret |= 2;
        return ret;
    }
    case ORC_CNST_QAM16: {
        /* GR's if/else range chain, first match wins => strict comparisons below */
        const float level = sqrtf((float)0.1);
        unsigned ret = 0;
        if (s.im > 0) ret |= 8;
        if (s.re < 0) ret |= 4;
        if (fabsf(s.re) > 2 * level) ret |= 2;
        if (fabsf(s.im) > 2 * level) ret |= 1;
        return ret;
    }
    default: return 0;
    }
}

/* GR3.10 constellation::calc_soft_dec(sample, npwr): for every point d = expf(-dist/npwr),
 * accumulated per bit into P(0)/P(1); LLR_j = logf(P1) - logf(P0); returned MSB first.
 * (d_scalefactor = 1 for the built-in constellations.) */
/* PRODUCTION ("fast") arithmetic for 16-QAM with squared distances (NOT the reference's bits): the GR 16-QAM
 * table is a 4 x 4 grid -- index bits 1, 2 select the real level, bits 0, 3 the imaginary level
 * (orc_decision_maker) -- and exp(-(dx^2 + dy^2)/npwr) = exp(-dx^2/npwr) exp(-dy^2/npwr), so the sums over the 8
 * points with bit j set / clear share the other axis' factor, which cancels in the ratio: P(bit) needs only the 4
 * levels of the bit's own axis.  8 exp + 8 log per symbol instead of 16 + 8.  Levels are visited in the order of the
 * points' indices (real: points 0, 2, 4, 6; imaginary: points 0, 1, 8, 9).  out: MSB first. */
static void calc_soft_dec_qam16_fast(orc_c32 sample, float npwr, float* out)
{
    orc_c32 pts[16];
    orc_constellation_points(ORC_CNST_QAM16, pts);
    float er[4], ei[4];
    for (int q = 0; q < 4; q++) {
        float dr = sample.re - pts[q << 1].re;                       /* real level of index bits (2,1) = q     */
        float di = sample.im - pts[(q & 1) | ((q >> 1) << 3)].im;    /* imaginary level of index bits (3,0) = q */
        er[q] = orc_expf(-(dr * dr) / npwr);
        ei[q] = orc_expf(-(di * di) / npwr);
    }
    /* bit 1 = LSB of the real level index, bit 2 = its MSB; bit 0 = LSB of the imaginary level index, bit 3 = MSB */
    float s[4][2];
    s[1][0] = er[0] + er[2]; s[1][1] = er[1] + er[3];
    s[2][0] = er[0] + er[1]; s[2][1] = er[2] + er[3];
    s[0][0] = ei[0] + ei[2]; s[0][1] = ei[1] + ei[3];
    s[3][0] = ei[0] + ei[1]; s[3][1] = ei[2] + ei[3];
    for (int i = 0; i < 4; i++) out[3 - i] = orc_logf(s[i][1]) - orc_logf(s[i][0]);
}

void orc_calc_soft_dec(int cnst, orc_c32 sample, float npwr, int dist_mode, float* out)
{
    if ((dist_mode & ORC_DEMAP_FAST) && cnst == ORC_CNST_QAM16 && (dist_mode & 0xff) == 0) {
        calc_soft_dec_qam16_fast(sample, npwr, out);
        return;
    }
    dist_mode &= 0xff;
    orc_c32 pts[16];
    int M = orc_constellation_points(cnst, pts);
    int k = orc_bits_per_symbol(cnst);
    float tmp[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < M; i++) {
        float dr = sample.re - pts[i].re, di = sample.im - pts[i].im;
        float dist = dist_mode ? sqrtf(dr * dr + di * di) : (dr * dr + di * di);
        float d = orc_expf(-dist / npwr); /* expf with a fixed operation sequence, orc_math.h */
        for (int j = 0; j < k; j++) {
            int bit = (i >> j) & 1;
            tmp[2 * j + bit] += d;
        }
    }
    for (int i = 0; i < k; i++) out[k - 1 - i] = orc_logf(tmp[2 * i + 1]) - orc_logf(tmp[2 * i + 0]);
}

/* lib/dtl/ofdm_adaptive_constellation_soft_cf_impl.cc:89-103 (sigma = sqrt(noise) or 0.001)
 * and :143-148 (calc_soft_dec(sample, sigma) then std::reverse => LSB first) */
void orc_soft_demap(int cnst, const orc_c32* syms, int n_syms, double noise, int dist_mode,
                    float* llr_out)
{
    double sigma = 0.001;
    if (!(noise < 0)) sigma = sqrt(noise); /* negative: no noise tag (soft_cf_impl.cc:89); a NaN estimate is a tag like any other */
    int bps = orc_bits_per_symbol(cnst);
    float msb[4];
    for (int i = 0; i < n_syms; i++) {
        orc_calc_soft_dec(cnst, syms[i], (float)sigma, dist_mode, msb);
        for (int j = 0; j < bps; j++) llr_out[i * bps + j] = msb[bps - 1 - j];
    }
}

/* test hooks: the fixed-sequence math of orc_math.h over arrays (tests/test_math_accuracy_cpu.py) */
void orc_math_logf(const float* x, long n, float* y) { for (long i = 0; i < n; i++) y[i] = orc_logf(x[i]); }
void orc_math_expf(const float* x, long n, float* y) { for (long i = 0; i < n; i++) y[i] = orc_expf(x[i]); }
void orc_math_sincos_phase(const uint32_t* a, long n, float* sn, float* cs) { for (long i = 0; i < n; i++) orc_sincos_phase(a[i], &sn[i], &cs[i]); }
