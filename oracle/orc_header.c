/*
 * orc_header.c -- ORACLE (test infrastructure, not product): packet header format/parse and the
 * MCS feedback decision.  Follows lib/dtl/ofdm_adaptive_packet_header.cc and
 * lib/dtl/ofdm_adaptive_feedback_decision.cc.  Pinned by the six golden header rows of
 * python/dtl/qa_ofdm_adaptive_packet_header.py:130-135,236-241 and the decision sequence of
 * python/dtl/qa_ofdm_adaptive_feedback_decision.py:50-59.
 */
#include "oracle.h"

#include <float.h>
#include <string.h>

/* add_header_field (:78-90) with bits_per_byte = 1, mask = 1 */
static int put_field(uint8_t* buf, int header_len, int k, unsigned long long val, int n_bits)
{
    for (int i = 0; i < n_bits && k < header_len; i++, k++) buf[k] = (uint8_t)((val >> i) & 1u);
    return k;
}

/* pack_crc (:93-105): MSB-first packing of the header bits before the CRC field, including
 * the reference's loop bound `i*len+j < d_header_len` (for the 96-bit header the last byte
 * only takes 6 bits). */
static int pack_crc(const uint8_t* bits, int header_len, uint8_t* crc_buf)
{
    int len = (header_len - 16) / 8;
    if ((header_len - 16) % 8) ++len;
    memset(crc_buf, 0, len);
    for (int i = 0; i < len; ++i)
        for (int j = 0; j < 8 && i * len + j < header_len; j++)
            crc_buf[i] = (uint8_t)((crc_buf[i] << 1) | (bits[i * 8 + j] & 0x01));
    return len;
}

/* GR3.10 packet_header_ofdm ctor: scramble mask from lfsr(0x8a, 0x6f, 7), bits_per_sym = 1 */
static void scramble_mask(int header_len, uint8_t* mask)
{
    uint64_t reg = 0x6f;
    for (int i = 0; i < header_len; i++) {
        uint8_t out = (uint8_t)(reg & 1);
        uint64_t newbit = (uint64_t)(__builtin_popcountll(reg & 0x8a) % 2);
        reg = (reg >> 1) | (newbit << 7);
        mask[i] = out;
    }
}

static unsigned crc16_header(const uint8_t* bits, int header_len)
{
    uint8_t buf[1024]; /* header_len <= 8192 bits */
    int len = pack_crc(bits, header_len, buf);
    /* d_crc(16, 0x1021, 0xFFFF, 0, false, true), :72 */
    return (unsigned)orc_crc_compute(16, 0x1021, 0xFFFF, 0, 0, 1, buf, (size_t)len);
}

void orc_header_format(const orc_header* h, int has_fec, int header_len, int scramble,
                       uint8_t* out)
{
    memset(out, 0, header_len);
    int k = 0;
    k = put_field(out, header_len, k, (unsigned)h->payload_len & 0xFFF, 12);
    k = put_field(out, header_len, k, (unsigned)h->frame_no & 0xFFF, 12);
    k = put_field(out, header_len, k, (unsigned)h->cnst & 0xF, 4);
    k = put_field(out, header_len, k, (unsigned)h->fb_cnst & 0xF, 4);
    if (has_fec) { /* add_fec_header (:108-138): offsets 0,12,16,28,32 from k */
        put_field(out, header_len, k + 0, (unsigned)h->tb_no, 12);
        put_field(out, header_len, k + 12, (unsigned)h->fec_fb, 4);
        put_field(out, header_len, k + 16, (unsigned)h->tb_offset, 12);
        put_field(out, header_len, k + 28, (unsigned)h->fec, 4);
        put_field(out, header_len, k + 32, (unsigned)h->tb_payload, 16);
        k += 48;
    }
    unsigned crc = crc16_header(out, header_len);
    put_field(out, header_len, k, crc, 16);
    if (scramble) {
        uint8_t mask[8192];
        scramble_mask(header_len, mask);
        for (int i = 0; i < header_len; i++) out[i] ^= mask[i];
    }
}

static int get_field(const uint8_t* in, int header_len, int* k, int n_bits)
{
    int val = 0;
    for (int i = 0; i < n_bits && *k < header_len; i++, (*k)++) val |= (in[*k] & 1) << i;
    return val;
}

int orc_header_parse(const uint8_t* in_raw, int has_fec, int header_len, int scramble,
                     int* state_cnst, orc_header* h, int* n_syms)
{
    /* NB the reference's parser reads `in` as is: it does NOT undo the scramble mask that
     * header_formatter applies (:194-197 vs :231-312); scramble_header defaults to False
     * (python/dtl/ofdm_adaptive_config.py:38).  Restated faithfully: no descrambling. */
    const uint8_t* in = in_raw;
    (void)scramble;
    memset(h, 0, sizeof(*h));
    int k = 0;
    h->payload_len = get_field(in, header_len, &k, 12);
    h->frame_no = get_field(in, header_len, &k, 12);
    h->cnst = get_field(in, header_len, &k, 4);
    h->fb_cnst = get_field(in, header_len, &k, 4);
    if (has_fec) { /* parse_fec_header (:202-229) */
        h->tb_no = get_field(in, header_len, &k, 12);
        h->fec_fb = get_field(in, header_len, &k, 4);
        h->tb_offset = get_field(in, header_len, &k, 12);
        h->fec = get_field(in, header_len, &k, 4);
        h->tb_payload = get_field(in, header_len, &k, 16);
    }
    unsigned crc = crc16_header(in, header_len);
    for (int i = 0; i < 16 && k < header_len; i++, k++)
        if ((unsigned)(in[k] & 1) != ((crc >> i) & 1u)) return 0;
    /* :269-273 keep the previous MCS if the field is not 1..4 */
    if (h->cnst && h->cnst <= ORC_CNST_QAM16) *state_cnst = h->cnst;
    int bps = orc_bits_per_symbol(*state_cnst);
    if (bps == 0) return 0;
    int syms = h->payload_len * 8 / bps;
    if (h->payload_len * 8 % bps) syms++;
    *n_syms = syms;
    h->cnst = *state_cnst; /* the constellation tag carries d_constellation (:297-299) */
    return 1;
}

/* ---------------------------------------------------------------- feedback decision */

void orc_feedback_init(orc_feedback* f, double hysteresis, int decision_th, int n_mcs,
                       const double* snr_th, const int* cnst, const int* fec, int initial_mcs)
{
    memset(f, 0, sizeof(*f));
    f->n_mcs = n_mcs;
    for (int i = 0; i < n_mcs && i < 16; i++) {
        f->snr_th[i] = snr_th[i];
        f->cnst[i] = cnst[i];
        f->fec[i] = fec[i];
    }
    f->snr_th[0] = DBL_MIN; /* :50 */
    f->hysteresis = hysteresis;
    f->decision_th = decision_th;
    f->last_decision = f->new_decision = initial_mcs;
}

static void update_decision(orc_feedback* f, int mcs_id)
{ /* :82-93 */
    if (mcs_id != f->new_decision) {
        f->decision_counter = 0;
        f->new_decision = mcs_id;
    } else if (++f->decision_counter >= f->decision_th) {
        f->decision_counter = 0;
        f->last_decision = f->new_decision;
    }
}

void orc_feedback_get(orc_feedback* f, double snr, int* cnst, int* fec)
{ /* :56-79 */
    int cur = f->last_decision;
    if (snr < f->snr_th[cur]) {
        /* :66 `assert(current_mcs_id > 0)`: with snr_th[0] = DBL_MIN a negative-dB estimate at
         * MCS 0 is undefined behaviour in the reference; restated as "stay at MCS 0". */
        update_decision(f, cur > 0 ? cur - 1 : 0);
    } else if (cur + 1 < f->n_mcs) {
        if (snr > f->snr_th[cur + 1] + f->hysteresis) update_decision(f, cur + 1);
        else f->decision_counter = 0;
    } else {
        f->decision_counter = 0;
    }
    *cnst = f->cnst[f->last_decision];
    *fec = f->fec[f->last_decision];
}
