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
void orc_calc_soft_dec(int cnst, orc_c32 sample, float npwr, int dist_mode, float* out)
{
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
    if (noise >= 0) sigma = sqrt(noise);
    int bps = orc_bits_per_symbol(cnst);
    float msb[4];
    for (int i = 0; i < n_syms; i++) {
        orc_calc_soft_dec(cnst, syms[i], (float)sigma, dist_mode, msb);
        for (int j = 0; j < bps; j++) llr_out[i * bps + j] = msb[bps - 1 - j];
    }
}
