/*
 * orc_tb.c -- ORACLE (test infrastructure, not product): transport-block (TB) encode/decode and
 * FEC framing.  Follows lib/dtl/fec_utils.cc, lib/dtl/tb_encoder.cc, lib/dtl/tb_decoder.cc,
 * lib/dtl/ofdm_adaptive_fec_frame_bvb_impl.cc (TX, test-vector generator) and
 * lib/dtl/ofdm_adaptive_fec_decoder_impl.cc (RX).
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* lib/dtl/fec_utils.cc:104-112 */
int orc_compute_tb_len(int cw_len, int frame_len)
{
    int ncws = 1;
    if (frame_len > cw_len) ncws = 1 + frame_len / cw_len;
    return ncws;
}

/* lib/dtl/fec_utils.cc:114-121 */
int orc_align_bits_to_bytes(int nbits)
{
    int nbytes = nbits / 8;
    if (nbits % 8) ++nbytes;
    return nbytes;
}

/* lib/dtl/tb_encoder.cc:32-74 */
int orc_tb_encode(const orc_ldpc* c, const uint8_t* in, int len, int ncws, uint8_t* tb)
{
    const int n = orc_ldpc_n(c), k = orc_ldpc_k(c), ncheck = n - k;
    uint8_t* cw_in = (uint8_t*)malloc(n);
    uint8_t* cw_out = (uint8_t*)malloc(n);
    int payload = 0, read_index = 0, w = 0;
    for (int i = 0; i < ncws; ++i) {
        int k_new = (len - payload) / (ncws - i);
        if ((len - payload) % (ncws - i)) ++k_new;
        payload += k_new;
        memset(cw_in, 0, n);
        memcpy(cw_in, &in[read_index], k_new);
        orc_ldpc_encode(c, cw_in, cw_out);
        read_index += k_new;
        memcpy(&tb[w], cw_out, ncheck); w += ncheck;
        memcpy(&tb[w], &cw_out[ncheck], k_new); w += k_new;
    }
    free(cw_in); free(cw_out);
    return w;
}

static int align_bytes_to_syms(int nbytes, int bps)
{ /* ofdm_adaptive_fec_frame_bvb_impl.cc:247-254 */
    int nsyms = nbytes * 8 / bps;
    if (nbytes * 8 % bps) ++nsyms;
    return nsyms;
}

/* Steady-state path of ofdm_adaptive_fec_frame_bvb_impl::general_work (:284-565): one large
 * PDU (pdu_consumer.cc:45-52 takes tb_payload_max bytes per TB), unlimited output space, no
 * pacing, constant MCS.  Action PROCESS_INPUT -> OUTPUT_BUFFER -> FINALIZE_FRAME. */
int orc_fec_frames_build(const orc_ldpc* c, int fec_idx, int cnst, int cap, const uint8_t* user,
                         int n_user, int first_frame_no, uint8_t* syms_out, orc_header* hdr_out,
                         int max_frames)
{
    const int n = orc_ldpc_n(c), k = orc_ldpc_k(c);
    const int bps = orc_bits_per_symbol(cnst);
    const int frame_len_bytes = cap * bps / 8;                      /* :352 */
    const int ncws = orc_compute_tb_len(n, frame_len_bytes * 8);    /* :353 */
    const int tb_payload_max = (ncws * k) / 8 - 4;                  /* :360-361 */
    if (tb_payload_max <= 0) return -1;
    uint8_t* crc_buf = (uint8_t*)malloc(tb_payload_max + 8);
    uint8_t* tb_bits_in = (uint8_t*)malloc((size_t)ncws * k + 64);
    uint8_t* tb = (uint8_t*)malloc((size_t)ncws * n + 64);

    int n_frames = 0, frame_no = first_frame_no;
    int frame_offset = 0, used_capacity = 0, frame_payload = 0; /* d_current_frame_offset, d_frame_used_capacity, d_current_frame_payload */
    int write_pos = 0;                                          /* write_index inside the current frame */
    unsigned long tb_count = 0;
    int tb_payload_bits = 0;
    int read = 0;
    uint8_t* cur = &syms_out[0];
    memset(cur, 0, cap);

#define EMIT_FRAME(payload_bytes)                                                              \
    do {                                                                                       \
        if (n_frames >= max_frames) { n_frames = -2; goto done; }                              \
        orc_header* h = &hdr_out[n_frames];                                                    \
        memset(h, 0, sizeof(*h));                                                              \
        h->payload_len = (payload_bytes);            /* payload_length_key                */  \
        h->frame_no = frame_no & 0xFFF;                                                        \
        h->cnst = cnst; h->fb_cnst = 0; h->fec_fb = 0;                                         \
        h->tb_no = (int)(tb_count & 0xff);           /* :213 fec_tb_key                   */  \
        h->tb_offset = frame_offset & 0xfff;         /* :209-211 fec_offset_key           */  \
        h->fec = fec_idx;                                                                      \
        h->tb_payload = tb_payload_bits & 0xffff;    /* :204-207 fec_tb_payload_key       */  \
        frame_no = (frame_no + 1) & 0xFFF;                                                     \
        n_frames++;                                                                            \
        cur = &syms_out[(size_t)n_frames * cap];                                               \
        if (n_frames < max_frames) memset(cur, 0, cap);                                        \
        frame_offset = 0; frame_payload = 0; used_capacity = 0; write_pos = 0;                 \
    } while (0)

    while (read < n_user) {
        /* PROCESS_INPUT (:340-411) */
        int to_read = n_user - read < tb_payload_max ? n_user - read : tb_payload_max;
        memcpy(crc_buf, &user[read], to_read);
        orc_crc32_append(crc_buf, to_read);
        orc_repack_lsb_first(crc_buf, to_read + 4, tb_bits_in, 8, 1, 0);
        int tb_size = orc_tb_encode(c, tb_bits_in, (to_read + 4) * 8, ncws, tb);
        tb_payload_bits = (to_read + 4) * 8;
        read += to_read;
        ++tb_count;
        int used_frames_count = 0, buf_idx = 0;
        /* OUTPUT_BUFFER (:416-470) */
        while (buf_idx < tb_size) {
            int remaining = tb_size - buf_idx;
            int bytes_left = orc_align_bits_to_bytes(remaining);
            int avail = (cap - used_capacity) * bps / 8;
            int out_frame_bytes = bytes_left < avail ? bytes_left : avail;
            if (n_frames >= max_frames) { n_frames = -2; goto done; }
            memset(&cur[write_pos], 0, cap - write_pos);
            int nbits = out_frame_bytes * 8 < remaining ? out_frame_bytes * 8 : remaining;
            orc_repack_lsb_first(&tb[buf_idx], nbits, &cur[write_pos], 1, bps, 0); /* tb_encoder.cc:88-99 */
            buf_idx += nbits;
            if (out_frame_bytes == avail) {
                int payload_bytes = used_capacity * bps / 8 + out_frame_bytes;
                EMIT_FRAME(payload_bytes);
            } else if (frame_offset == 0) {
                frame_offset = out_frame_bytes;
                frame_payload += out_frame_bytes;
                used_capacity = align_bytes_to_syms(out_frame_bytes, bps);
                write_pos += used_capacity;
            } else {
                used_capacity = frame_offset + align_bytes_to_syms(out_frame_bytes, bps); /* sic, :459-460 */
                frame_payload += out_frame_bytes;
                write_pos += used_capacity;
            }
            ++used_frames_count;
        }
        /* :473-489 */
        if (used_frames_count == 1 || read == n_user) {
            /* FINALIZE_FRAME (:498-520), only if something is pending in the frame */
            if (frame_payload > 0 || used_capacity > 0) EMIT_FRAME(frame_payload);
        }
    }
done:
#undef EMIT_FRAME
    free(crc_buf); free(tb_bits_in); free(tb);
    return n_frames;
}

/* ------------------------------------------------------------------ RX: tb_decoder */

struct orc_tbdec {
    const orc_ldpc* code;
    int frame_capacity, max_bps, max_it;
    float bp_sigma;
    int arith;
    /* tb_decoder members (tb_decoder.h:29-38) */
    float* rcv;  long rcv_size, rcv_cap;     /* d_tb_buffers[RCV_BUF]; stale contents persist */
    float* full;                             /* d_tb_buffers[FULL_BUF]                        */
    uint8_t* data; int data_len;             /* d_data_buffer                                 */
    int tb_number;                           /* d_tb_number, -1 at start                      */
    long buf_idx;
    int have_info, info_tb_payload;          /* d_fec_info (only the fields decode() uses)    */
    uint8_t* crc_buffer;
    long data_cap;                           /* bits d->data / d->crc_buffer can hold (vectors in the reference) */
    long crc_ok_count, crc_fail_count;
    int* nit_log; long nit_cap; long* nit_count;   /* optional: iteration count of every codeword, in decode order */
};

void orc_tbdec_set_nit_log(orc_tbdec* d, int* log, long cap, long* count)
{
    d->nit_log = log; d->nit_cap = cap; d->nit_count = count;
}

orc_tbdec* orc_tbdec_new(const orc_ldpc* c, int frame_capacity, int max_bps, int max_it,
                         float bp_sigma)
{
    orc_tbdec* d = (orc_tbdec*)calloc(1, sizeof(*d));
    d->code = c; d->frame_capacity = frame_capacity; d->max_bps = max_bps;
    d->max_it = max_it; d->bp_sigma = bp_sigma;
    int n = orc_ldpc_n(c), k = orc_ldpc_k(c);
    int frame_len = frame_capacity * max_bps;                 /* fec_decoder_impl.cc:68 */
    int ncws = orc_compute_tb_len(n, frame_len);
    int max_tb_len = n * ncws;
    d->rcv_cap = 2L * max_tb_len;                              /* tb_decoder.cc:24-27 */
    d->rcv = (float*)calloc(d->rcv_cap + 64, sizeof(float));
    d->full = (float*)calloc(2L * max_tb_len + 64, sizeof(float));
    d->data_cap = 2L * max_tb_len;
    d->data = (uint8_t*)calloc(d->data_cap + k + 64, 1);
    d->crc_buffer = (uint8_t*)calloc(d->data_cap / 8 + 16, 1);
    d->tb_number = -1;
    return d;
}

void orc_tbdec_set_arith(orc_tbdec* d, int arith) { d->arith = arith; }

void orc_tbdec_free(orc_tbdec* d)
{
    if (!d) return;
    free(d->rcv); free(d->full); free(d->data); free(d->crc_buffer); free(d);
}

static void rcv_append(orc_tbdec* d, const float* in, long lo, long hi, long in_len)
{ /* std::copy(in+lo, in+hi, back_inserter(RCV)); reads past the frame are taken as 0 */
    for (long i = lo; i < hi; i++) {
        float v = (i < in_len) ? in[i] : 0.0f;
        if (d->rcv_size < d->rcv_cap + 64) d->rcv[d->rcv_size] = v;
        d->rcv_size++;
    }
}

/* tb_decoder::decode (tb_decoder.cc:143-195) */
static void tb_decode(orc_tbdec* d, int tb_len, int* avg_it)
{
    const int n = orc_ldpc_n(d->code), k = orc_ldpc_k(d->code), ncheck = n - k;
    int payload_len = d->info_tb_payload;
    d->data_len = d->info_tb_payload;
    if (d->info_tb_payload > d->data_cap) { /* d_data_buffer.resize(): a header may announce any 16-bit length */
        d->data_cap = d->info_tb_payload;
        d->data = (uint8_t*)realloc(d->data, d->data_cap + k + 64);
        d->crc_buffer = (uint8_t*)realloc(d->crc_buffer, d->data_cap / 8 + 16);
    }
    for (long i = 0; i < (long)tb_len * n; i++) d->full[i] = -15.0f; /* SHORTENED_VALUE */
    long rcv_idx = 0;
    int d_idx = 0, sum_it = 0;
    uint8_t* cw_out = (uint8_t*)malloc(k);
    for (int i = 0; i < tb_len; ++i) {
        int k_ = payload_len / (tb_len - i);
        if (payload_len % (tb_len - i)) ++k_;
        payload_len -= k_;
        /* The reference copies from RCV_BUF past its size() when frames of the TB were lost
         * (tb_decoder.cc:175-182): undefined behaviour.  Restated as "soft bits that were never
         * received carry no information": value 0. */
        for (int j = 0; j < ncheck; j++, rcv_idx++)
            d->full[(long)i * n + j] = (rcv_idx < d->rcv_size) ? d->rcv[rcv_idx] : 0.0f;
        /* k_ > k (see below): the reference's copy runs on into the next codeword's slots, which that codeword's
         * own copy then overwrites (k_ - k <= ncheck + its k_) -- never seen by a decode, so not stored here */
        for (int j = 0; j < k_; j++, rcv_idx++)
            if (j < k) d->full[(long)i * n + ncheck + j] = (rcv_idx < d->rcv_size) ? d->rcv[rcv_idx] : 0.0f;
        int nit = 0;
        orc_ldpc_decode_arith(d->code, &d->full[(long)i * n], d->max_it, d->bp_sigma, d->arith, cw_out, &nit);
        /* the reference decodes k bits straight into its data buffer at d_idx and advances by k_: for k_ < k the
         * next codeword overwrites the tail; for k_ > k (a block flushed with the next frame's constellation,
         * tb_decoder.cc:99-103) bits d_idx+k .. d_idx+k_ keep what an earlier block left there.  Restated as 0:
         * the block's CRC-32 fails either way */
        memcpy(&d->data[d_idx], cw_out, k_ < k ? k_ : k);
        if (k_ > k) memset(&d->data[d_idx + k], 0, k_ - k);
        d_idx += k_;
        sum_it += nit;
        if (d->nit_log && d->nit_count) {
            if (*d->nit_count < d->nit_cap) d->nit_log[*d->nit_count] = nit;
            (*d->nit_count)++;
        }
    }
    free(cw_out);
    *avg_it = sum_it / tb_len;
}

struct emit_ctx {
    orc_tbdec* d; uint8_t* out; long out_cap; long* out_len; orc_tb_rec* recs; int recs_cap;
    int* n_recs;
};

/* on_data_ready lambda (ofdm_adaptive_fec_decoder_impl.cc:161-201) */
static void on_data_ready(struct emit_ctx* e, int tb_no, int tb_payload, int avg_it)
{
    orc_tbdec* d = e->d;
    int user_data_len = d->data_len - 32;
    int crc_ok = 0;
    if (user_data_len >= 0) {
        int crc_buf_len = orc_repack_lsb_first(d->data, user_data_len, d->crc_buffer, 1, 8, 0);
        orc_repack_lsb_first(&d->data[user_data_len], 32, &d->crc_buffer[crc_buf_len], 1, 8, 0);
        crc_ok = orc_crc32_verify(d->crc_buffer, crc_buf_len + 4);
    }
    if (crc_ok) d->crc_ok_count++; else d->crc_fail_count++;
    long off = -1;
    if (crc_ok && *e->out_len + user_data_len <= e->out_cap) {
        off = *e->out_len;
        memcpy(&e->out[off], d->data, user_data_len);
        *e->out_len += user_data_len;
    }
    if (*e->n_recs < e->recs_cap) {
        orc_tb_rec* r = &e->recs[*e->n_recs];
        r->tb_no = tb_no; r->tb_payload = tb_payload; r->crc_ok = crc_ok; r->avg_it = avg_it;
        r->user_bits = user_data_len; r->out_offset = off;
    }
    (*e->n_recs)++;
}

/* ofdm_adaptive_fec_decoder_impl::general_work (:80-217) for one frame, then
 * tb_decoder::process_frame (tb_decoder.cc:32-141) */
int orc_tbdec_frame(orc_tbdec* d, const float* llr, int len, int cnst, const orc_header* h,
                    uint8_t* out, long out_cap, long* out_len, orc_tb_rec* recs, int recs_cap,
                    int* n_recs)
{
    const int n = orc_ldpc_n(d->code), k = orc_ldpc_k(d->code), ncheck = n - k;
    const int bps = orc_bits_per_symbol(cnst);
    if (bps == 0) return -4; /* "Tags missing" */
    const int frame_payload = 8 * h->payload_len;            /* fec_utils.cc:93 */
    if (frame_payload == 0) return 0;                        /* :141-143 empty frame */
    const int tb_offset = h->tb_offset * 8;                  /* :147 */
    const int frame_len = 8 * orc_align_bits_to_bytes(d->frame_capacity * bps); /* :207 */
    int before = *n_recs, avg_it = 0;
    struct emit_ctx e = { d, out, out_cap, out_len, recs, recs_cap, n_recs };

    if (h->tb_no == d->tb_number) {
        /* frame is part of the current TB (:56-70) */
        if (d->buf_idx + frame_payload > d->rcv_cap) return -5; /* "rcv buffer does not have enough capacity!" */
        rcv_append(d, llr, 0, frame_payload, len);
        d->buf_idx += frame_payload;
        int ncws = orc_compute_tb_len(n, frame_len);
        if (d->rcv_size >= (long)ncheck * ncws + h->tb_payload) { /* expected_tb_len(fec_info, ncws) */
            tb_decode(d, ncws, &avg_it);
            d->buf_idx = 0;
            d->rcv_size = 0;
            on_data_ready(&e, h->tb_no, h->tb_payload, avg_it);
        }
    } else if (d->rcv_size == 0 && tb_offset == frame_payload) {
        /* small TB exclusively transported by the frame (:74-89) */
        d->have_info = 1; d->info_tb_payload = h->tb_payload;
        d->tb_number = h->tb_no;
        d->buf_idx = 0; d->rcv_size = 0;
        rcv_append(d, llr, 0, tb_offset, len);
        int tb_len = orc_compute_tb_len(n, frame_len);
        tb_decode(d, tb_len, &avg_it);
        d->rcv_size = 0;
        on_data_ready(&e, h->tb_no, h->tb_payload, avg_it);
    } else {
        /* finish the current TB with the first tb_offset soft bits (:92-104) */
        if (tb_offset > 0) rcv_append(d, llr, 0, tb_offset, len);
        if (d->rcv_size > 0 && d->have_info) {
            int tb_len = orc_compute_tb_len(n, frame_len);
            int old_no = d->tb_number, old_payload = d->info_tb_payload;
            tb_decode(d, tb_len, &avg_it);
            on_data_ready(&e, old_no, old_payload, avg_it);
        }
        /* start the new TB (:106-135) */
        d->have_info = 1; d->info_tb_payload = h->tb_payload;
        d->tb_number = h->tb_no;
        d->buf_idx = 0; d->rcv_size = 0;
        int extra_bits = 0;
        if (tb_offset % bps) extra_bits = bps - tb_offset % bps;
        int new_tb_offset = 0;
        if (tb_offset) new_tb_offset = tb_offset + extra_bits;
        rcv_append(d, llr, new_tb_offset, (long)frame_payload + extra_bits, len);
        d->buf_idx += frame_payload - tb_offset;
        int tb_len = orc_compute_tb_len(n, frame_len);
        int tb_size = 8 * orc_align_bits_to_bytes(tb_len * ncheck + h->tb_payload);
        if (frame_payload - tb_offset == tb_size) {
            tb_decode(d, tb_len, &avg_it);
            on_data_ready(&e, h->tb_no, h->tb_payload, avg_it);
            d->buf_idx = 0; d->rcv_size = 0;
        }
    }
    return *n_recs - before;
}

/* lib/dtl/ofdm_adaptive_fec_pack_bb_impl.cc:41-62 */
long orc_fec_pack(const uint8_t* bits, long nbits, uint8_t* bytes)
{
    long to_pack = 8 * (nbits / 8);
    return orc_repack_lsb_first(bits, (size_t)to_pack, bytes, 1, 8, 1);
}
