/*
 * orc_bits.c -- ORACLE (test infrastructure, not product): bit repacking and CRCs.
 * Follows lib/testbed/repack.cc and lib/dtl/crc_util.cc; gr::digital::crc is GR 3.10
 * (restated; pinned by the header CRC-16 golden vectors and by the standard CRC-32 check
 * value 0xCBF43926 for "123456789").
 */
#include "oracle.h"

/* lib/testbed/repack.cc:31-69 */
int orc_repack_lsb_first(const uint8_t* in, size_t n_in, uint8_t* out, int bits_in,
                         int bits_out, int full_output_symbols)
{
    unsigned in_index = 0, out_index = 0;
    size_t bytes_to_write = n_in * bits_in / bits_out;
    if (((n_in * bits_in) % bits_out) != 0) bytes_to_write += (size_t)(!full_output_symbols);
    size_t n_read = 0, n_written = 0;
    while (n_written < bytes_to_write && n_read < n_in) {
        if (out_index == 0) out[n_written] = 0;
        out[n_written] |= ((in[n_read] >> in_index) & 0x01) << out_index;
        in_index = (in_index + 1) % bits_in;
        out_index = (out_index + 1) % bits_out;
        if (in_index == 0) n_read++;
        if (out_index == 0) n_written++;
    }
    if (out_index) { n_written++; out_index = 0; }
    return (int)n_written;
}

/* lib/testbed/repack.cc:72-112 */
int orc_repack_msb_first(const uint8_t* in, size_t n_in, uint8_t* out, int bits_in,
                         int bits_out, int full_output_symbols)
{
    unsigned in_index = 0, out_index = 0;
    size_t bytes_to_write = n_in * bits_in / bits_out;
    if (((n_in * bits_in) % bits_out) != 0) bytes_to_write += (size_t)(!full_output_symbols);
    size_t n_read = 0, n_written = 0;
    while (n_written < bytes_to_write && n_read < n_in) {
        if (out_index == 0) out[n_written] = 0;
        out[n_written] |= ((in[n_read] >> (bits_in - 1 - in_index)) & 0x01)
                          << (bits_out - 1 - out_index);
        in_index = (in_index + 1) % bits_in;
        out_index = (out_index + 1) % bits_out;
        if (in_index == 0) n_read++;
        if (out_index == 0) n_written++;
    }
    if (out_index) { n_written++; out_index = 0; }
    return (int)n_written;
}

static uint64_t reflect(uint64_t v, int nbits)
{
    uint64_t r = 0;
    for (int i = 0; i < nbits; i++)
        if (v & (1ull << i)) r |= 1ull << (nbits - 1 - i);
    return r;
}

/* gr::digital::crc::compute (GR3.10), bit-serial form */
uint64_t orc_crc_compute(int num_bits, uint64_t poly, uint64_t init, uint64_t final_xor,
                         int input_reflected, int result_reflected, const uint8_t* data,
                         size_t len)
{
    const uint64_t mask = (num_bits == 64) ? ~0ull : ((1ull << num_bits) - 1);
    const uint64_t top = 1ull << (num_bits - 1);
    uint64_t reg = init & mask;
    for (size_t i = 0; i < len; i++) {
        uint8_t byte = input_reflected ? (uint8_t)reflect(data[i], 8) : data[i];
        for (int b = 7; b >= 0; b--) {
            uint64_t inbit = (byte >> b) & 1u;
            uint64_t msb = (reg & top) ? 1u : 0u;
            reg = (reg << 1) & mask;
            if (msb ^ inbit) reg ^= poly;
        }
    }
    if (result_reflected) reg = reflect(reg, num_bits);
    return (reg ^ final_xor) & mask;
}

static uint32_t crc32_std(const uint8_t* buf, size_t len)
{
    /* lib/dtl/crc_util.cc:23 : crc(32, poly, init, xorout, true, true) with the
     * arguments of ofdm_adaptive_fec_decoder_impl.cc:54 (0x04C11DB7, ~0, ~0) */
    return (uint32_t)orc_crc_compute(32, 0x04C11DB7ull, 0xFFFFFFFFull, 0xFFFFFFFFull, 1, 1, buf, len);
}

/* lib/dtl/crc_util.cc:29-36 */
uint32_t orc_crc32_append(uint8_t* buf, size_t len)
{
    uint32_t crc = crc32_std(buf, len);
    for (unsigned i = 0; i < 4; ++i) buf[len + i] = (crc >> (i * 8)) & 0xFF;
    return crc;
}

/* lib/dtl/crc_util.cc:41-61 (monitoring counters left to the caller) */
int orc_crc32_verify(const uint8_t* buf, size_t len)
{
    if (len < 4) return 0;
    size_t payload_len = len - 4;
    uint32_t crc = crc32_std(buf, payload_len);
    int ok = 1;
    for (unsigned i = 0; i < 4; ++i)
        if (buf[payload_len + i] != ((crc >> (i * 8)) & 0xFF)) ok = 0;
    return ok;
}
