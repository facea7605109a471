/*
 * orc_rx.c -- ORACLE (test infrastructure, not product): the whole receive graph of
 * python/dtl/ofdm_receiver.py:59-246 run over one sample stream, block by block, with the
 * header -> demux message loop of GR3.10 header_payload_demux (ofdm_receiver.py:79-86,150)
 * restated as a sequential cut loop.  Also the no-FEC frame builder used by config #1.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* lib/dtl/ofdm_adaptive_frame_bb_impl.cc:131-163 (frame_out) for one frame: user bytes +
 * CRC-32 + padding (the reference pads with random bytes; zeros here, not covered by the
 * CRC) repacked 8 -> bps LSB first.  Returns number of symbols (== frame_capacity). */
int orc_nofec_frame_build(int cnst, int frame_capacity, const uint8_t* user, int n_user,
                          int frame_no, uint8_t* syms_out, orc_header* h);

int orc_nofec_frame_build(int cnst, int cap, const uint8_t* user, int n_user, int frame_no,
                          uint8_t* syms_out, orc_header* h)
{
    int bps = orc_bits_per_symbol(cnst);
    int frame_in_bytes = cap * bps / 8 - 4; /* :236-237 */
    if (n_user > frame_in_bytes) n_user = frame_in_bytes;
    uint8_t* buf = (uint8_t*)calloc((size_t)frame_in_bytes + 8, 1);
    memcpy(buf, user, n_user);
    orc_crc32_append(buf, n_user);
    int nsyms = orc_repack_lsb_first(buf, frame_in_bytes + 4, syms_out, 8, bps, 0);
    memset(h, 0, sizeof(*h));
    h->payload_len = n_user ? n_user + 4 : 0; /* add_tags :330-343 */
    h->frame_no = frame_no & 0xFFF;
    h->cnst = cnst;
    free(buf);
    return nsyms;
}

int orc_rx_stream(const orc_cfg* cfg, const orc_ldpc* code, const orc_feedback* fb_init,
                  const orc_c32* iq, long n, orc_rx_out* o)
{
    const int N = cfg->fft_len, cp = cfg->cp_len, P = N + cp;
    const int h = cfg->header_len, L = cfg->frame_length, hdr_syms = 2 + h;
    const int cap = orc_frame_capacity(cfg);
    /* the formatter is built from occupied_carriers[0] alone, once per header symbol (python/dtl/ofdm_receiver.py:129-130);
     * the serializer delivers the header symbols' own sets (symbol i: set i mod n_sets): where the two differ the
     * reference's parser and demux fall out of step -- such a plan is refused (the product refuses it at create) */
    const int hdr_bits_len = h * cfg->occ_len[0];
    int hdr_carriers = 0;
    for (int i = 0; i < h; i++) hdr_carriers += cfg->occ_len[i % cfg->n_occ_sets];
    if (hdr_carriers != hdr_bits_len) return -3;
    if (hdr_bits_len < (cfg->has_fec ? 96 : 48)) return -3; /* the header fields do not fit: nothing would decode */

    uint8_t* trig_raw = (uint8_t*)malloc(n);
    uint8_t* trig_fix = (uint8_t*)malloc(n);
    float* angle = (float*)malloc(n * sizeof(float));
    orc_c32* mixed = (orc_c32*)malloc(n * sizeof(orc_c32));
    orc_sync_sc(cfg, iq, n, NULL, trig_raw, angle);
    if (cfg->use_sync_correct) {
        orc_framedetect fd;
        orc_framedetect_init(&fd, (L + hdr_syms) * P); /* ofdm_receiver.py:77-78 */
        orc_fix_sync(&fd, trig_raw, trig_fix, (int)n);
    } else {
        memcpy(trig_fix, trig_raw, n);
    }
    orc_cfo_mix_trig(cfg, iq, n, angle, trig_raw, mixed);
    if (o->trig_raw) memcpy(o->trig_raw, trig_raw, n);
    if (o->trig_fixed) memcpy(o->trig_fixed, trig_fix, n);
    if (o->angle) memcpy(o->angle, angle, n * sizeof(float));

    orc_c32* sym_t = (orc_c32*)malloc((size_t)N * (hdr_syms + L) * sizeof(orc_c32));
    orc_c32* sym_f = (orc_c32*)malloc((size_t)N * (hdr_syms + L) * sizeof(orc_c32));
    const int eq_syms = L > h ? L : h;  /* the header symbols go through the same buffers */
    orc_c32* hard = (orc_c32*)malloc((size_t)N * eq_syms * sizeof(orc_c32));
    orc_c32* soft = (orc_c32*)malloc((size_t)N * eq_syms * sizeof(orc_c32));
    orc_c32* ser = (orc_c32*)malloc((size_t)(cap + N) * sizeof(orc_c32));
    orc_c32* taps = (orc_c32*)malloc((size_t)N * sizeof(orc_c32));
    float* llr = (float*)malloc((size_t)cap * 4 * sizeof(float));
    uint8_t* bits = (uint8_t*)malloc((size_t)cap * 4 + 64);

    int state_cnst = ORC_CNST_QAM16; /* packet_header.cc:70 */
    orc_feedback fb = *fb_init;
    int expected_frame_no = 0;
    long lost_frames = 0;
    orc_tbdec* tbd = NULL;
    if (cfg->has_fec && code) { tbd = orc_tbdec_new(code, cap, 4, cfg->max_it, cfg->bp_sigma); orc_tbdec_set_arith(tbd, cfg->arith); }
    o->n_cw_nit = 0;
    if (tbd && o->cw_nit) orc_tbdec_set_nit_log(tbd, o->cw_nit, o->cw_nit_cap, &o->n_cw_nit);
    o->n_frames = 0; o->n_tbs = 0; o->n_user_bits = 0; o->n_user_bytes = 0;
    int rc = 0;

    long pos = 0;
    while (1) {
        long i = pos;
        while (i < n && !trig_fix[i]) i++;
        if (i >= n) break;
        if (i + (long)hdr_syms * P > n) break;
        /* header_payload_demux out0: strip CP, (2+h) symbols; fft_vcc */
        for (int s = 0; s < hdr_syms; s++) {
            memcpy(&sym_t[(size_t)s * N], &mixed[i + (long)s * P + cp], N * sizeof(orc_c32));
            orc_fft_shifted(&sym_t[(size_t)s * N], &sym_f[(size_t)s * N], N, 1);
        }
        int carr_offset = orc_chanest(cfg, sym_f, taps);
        /* digital.ofdm_frame_equalizer_vcvc(header_equalizer) -> serializer -> BPSK decoder */
        orc_frame_equalize(cfg, 1, ORC_CNST_BPSK, &sym_f[(size_t)2 * N], h, carr_offset, taps, hard, NULL, NULL);
        int nh = orc_serialize(cfg, hard, h, 0, 0, ser);
        for (int k = 0; k < nh; k++) bits[k] = (uint8_t)orc_decision_maker(ORC_CNST_BPSK, ser[k]);
        orc_header hdr;
        int n_payload_syms = 0;
        int ok = orc_header_parse(bits, cfg->has_fec, hdr_bits_len, cfg->scramble_header, &state_cnst, &hdr, &n_payload_syms);
        int fi = o->n_frames;
        if (fi >= o->max_frames) { rc = -2; break; }
        orc_frame_rec* rec = o->frames ? &o->frames[fi] : NULL;
        if (rec) { memset(rec, 0, sizeof(*rec)); rec->trigger = i; rec->header_ok = ok; rec->carr_offset = carr_offset; }
        if (o->hdr_bits) memcpy(&o->hdr_bits[(size_t)fi * hdr_bits_len], bits, hdr_bits_len);
        if (!ok) { /* header CRC failed: demux drops the frame and searches on (PMT_F) */
            o->n_frames++;
            pos = i + 1;
            continue;
        }
        if (i + (long)(hdr_syms + L) * P > n) break;
        if (o->taps) memcpy(&o->taps[(size_t)fi * N], taps, N * sizeof(orc_c32));
        /* header_payload_demux out1: L payload symbols; fft_vcc */
        for (int s = 0; s < L; s++) {
            memcpy(&sym_t[(size_t)s * N], &mixed[i + (long)(hdr_syms + s) * P + cp], N * sizeof(orc_c32));
            orc_fft_shifted(&sym_t[(size_t)s * N], &sym_f[(size_t)s * N], N, 1);
        }
        /* ofdm_adaptive_frame_equalizer_vcvc */
        orc_snr snr;
        orc_frame_equalize(cfg, 0, hdr.cnst, sym_f, L, carr_offset, taps, hard, soft, &snr);
        { /* lost-frame accounting :124-138 */
            int lost = 0;
            if (hdr.frame_no != expected_frame_no) {
                if (hdr.frame_no < expected_frame_no) lost = 4096 + hdr.frame_no - expected_frame_no;
                else lost = hdr.frame_no - expected_frame_no;
            }
            lost_frames += lost;
            expected_frame_no = (hdr.frame_no + 1) % 4096;
        }
        int fb_c, fb_f;
        orc_feedback_get(&fb, snr.snr_db, &fb_c, &fb_f);
        if (rec) {
            rec->hdr = hdr; rec->n_payload_syms = n_payload_syms; rec->snr_db = snr.snr_db;
            rec->noise_lin = snr.noise_lin; rec->fb_cnst = fb_c; rec->fb_fec = fb_f;
            rec->lost_frames = (int)lost_frames;
        }
        if (o->eq_hard) memcpy(&o->eq_hard[(size_t)fi * L * N], hard, (size_t)L * N * sizeof(orc_c32));
        if (o->eq_soft) memcpy(&o->eq_soft[(size_t)fi * L * N], soft, (size_t)L * N * sizeof(orc_c32));
        o->n_frames++;
        /* GR3.10 header_payload_demux_impl::general_work, STATE_PAYLOAD: "consume payload, except for the
         * last item" (it may hold the next trigger): consume_each(payload_len * (N + gi) - 1) */
        pos = i + (long)(hdr_syms + L) * P - 1;
        if (hdr.payload_len == 0) continue; /* work() returns 0: empty frame dropped (:218,254) */

        /* ofdm_serializer_vcc(..., packet_length key, symbols_skipped = 1) on the soft output */
        int ns = orc_serialize(cfg, soft, L, 1, n_payload_syms, ser);
        if (o->soft_syms) {
            memset(&o->soft_syms[(size_t)fi * cap], 0, (size_t)cap * sizeof(orc_c32));
            memcpy(&o->soft_syms[(size_t)fi * cap], ser, (size_t)ns * sizeof(orc_c32));
        }
        int bps = orc_bits_per_symbol(hdr.cnst);
        if (cfg->has_fec) {
            orc_soft_demap(hdr.cnst, ser, ns, snr.noise_lin, cfg->llr_dist_mode | (cfg->arith ? ORC_DEMAP_FAST : 0), llr);
            if (o->llr) {
                memset(&o->llr[(size_t)fi * cap * 4], 0, (size_t)cap * 4 * sizeof(float));
                memcpy(&o->llr[(size_t)fi * cap * 4], llr, (size_t)ns * bps * sizeof(float));
            }
            if (tbd && o->user_bits) {
                int nrec = o->n_tbs;
                int r = orc_tbdec_frame(tbd, llr, ns * bps, hdr.cnst, &hdr, o->user_bits, o->user_bits_cap,
                                        &o->n_user_bits, o->tbs, o->tbs_cap, &nrec);
                o->n_tbs = nrec;
                if (r < 0) { rc = r; break; }
            }
        } else {
            /* ofdm_adaptive_constellation_decoder_cb + ofdm_adaptive_frame_pack_bb */
            for (int k = 0; k < ns; k++) bits[k] = (uint8_t)orc_decision_maker(hdr.cnst, ser[k]);
            uint8_t* bytes = (uint8_t*)malloc((size_t)ns + 8);
            int nw = orc_repack_lsb_first(bits, ns, bytes, bps, 8, 0);
            if (ns * bps % 8) --nw;
            int crc_ok = orc_crc32_verify(bytes, nw);
            if (o->tbs && o->n_tbs < o->tbs_cap) {
                orc_tb_rec* r = &o->tbs[o->n_tbs];
                r->tb_no = hdr.frame_no; r->tb_payload = nw; r->crc_ok = crc_ok; r->avg_it = 0;
                r->user_bits = (nw - 4) * 8; r->out_offset = o->n_user_bytes;
            }
            o->n_tbs++;
            if (o->user_bytes && nw >= 4 && o->n_user_bytes + nw - 4 <= o->user_bytes_cap) {
                memcpy(&o->user_bytes[o->n_user_bytes], bytes, nw - 4);
                o->n_user_bytes += nw - 4;
            }
            free(bytes);
        }
    }
    if (cfg->has_fec && o->user_bytes && o->user_bits) /* ofdm_adaptive_fec_pack_bb */
        o->n_user_bytes = orc_fec_pack(o->user_bits, o->n_user_bits, o->user_bytes);

    orc_tbdec_free(tbd);
    free(trig_raw); free(trig_fix); free(angle); free(mixed); free(sym_t); free(sym_f);
    free(hard); free(soft); free(ser); free(taps); free(llr); free(bits);
    return rc;
}
