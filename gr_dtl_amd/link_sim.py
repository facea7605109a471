"""AWGN link simulation helpers around the product TX / RX: mixed-MCS streams and the bit-error accounting of a
decoded stream (the FEC-chain analogue of tools/ber.py: transport blocks are matched by their unwrapped 8-bit
number; blocks that were not delivered count as entirely wrong)."""
import numpy as np

ES = {1: 1.0, 2: 0.25, 3: 1.0, 4: 1.0}  # mean symbol energy per constellation (QPSK carries the 0.5 factor)


def mixed_mcs_stream(tx, segments, rng, fec_scheme="fec_1", lead_zeros=100, gap=0):
    """segments: [(constellation, n_user_bytes), ...] sent back to back with continuing frame / TB numbers.
    `gap` zero samples follow every frame (a paced transmitter; back-to-back frames make a trigger that is one
    sample early collide with the previous frame in the demux).
    -> (iq complex64 1-D with `lead_zeros` in front and one frame of zeros behind, sent TBs [(tb_no8, bytes)],
        mean payload-symbol energy of the stream)"""
    parts, tbs, energy, frame_no, tb_no, n_frames = [], [], 0.0, 0, 0, 0
    for cnst, nbytes in segments:
        user = rng.integers(0, 256, nbytes, dtype=np.uint8)
        iq, hdrs = tx.work(user, cnst, fec_scheme if tx.fec else None, first_frame_no=frame_no, first_tb_no=tb_no)
        if gap:
            iq = np.concatenate([iq, np.zeros((iq.shape[0], gap), np.complex64)], axis=1)
        parts.append(iq.reshape(-1))
        base_frame = frame_no
        frame_no = (frame_no + len(hdrs)) & 0xFFF
        energy += ES[int(cnst)] * len(hdrs)
        n_frames += len(hdrs)
        if tx.fec:
            enc = tx.ldpc_encs[tx.codes_id[fec_scheme]]
            bps = int(cnst) if int(cnst) < 4 else 4
            frame_bytes = tx.frame_capacity * bps // 8
            ncws = 1 + (frame_bytes * 8) // enc.get_n() if frame_bytes * 8 > enc.get_n() else 1
            per = ncws * enc.get_k() // 8 - 4
        else:
            per = tx.frame_capacity * (int(cnst) if int(cnst) < 4 else 4) // 8 - 4
        for lo in range(0, nbytes, per):
            tb_no += 1
            tbs.append((tb_no & 0xFF if tx.fec else (base_frame + lo // per) & 0xFFF, bytes(user[lo:lo + per])))
    x = np.concatenate([np.zeros(lead_zeros, np.complex64)] + parts + [np.zeros(tx.frame_samples, np.complex64)])
    return x, tbs, energy / max(n_frames, 1)


def add_awgn(x, noise_var, rng):
    """complex AWGN of total variance noise_var per sample"""
    s = np.sqrt(noise_var / 2.0)
    n = rng.standard_normal((x.size, 2)).astype(np.float32) * np.float32(s)
    return (x + n[:, 0] + 1j * n[:, 1]).astype(np.complex64)


def bit_errors(sent_tbs, rx_result, modulo):
    """Match delivered blocks (rx_result.tbs rows with crc_ok, bytes at out_offset) to the sent ones by number
    (mod `modulo`, in order) -> dict(bits_sent, bit_errors, blocks_sent, blocks_ok, ber, bler)."""
    payload = rx_result.payload_bytes().tobytes()
    bits_sent = sum(len(p) for _, p in sent_tbs) * 8
    errors, ok, cursor = 0, 0, 0
    delivered = set()
    for t in rx_result.tbs:
        if not t["crc_ok"] or t["out_offset"] < 0:
            continue
        nbytes = int(t["user_bits"]) // 8
        got = payload[int(t["out_offset"]):int(t["out_offset"]) + nbytes]
        # next sent block with this number at or after the cursor (numbers wrap, order does not)
        for j in range(cursor, min(cursor + modulo, len(sent_tbs))):
            if sent_tbs[j][0] == int(t["tb_no"]) % modulo and j not in delivered:
                want = sent_tbs[j][1]
                if len(want) == len(got):
                    errors += sum(bin(a ^ b).count("1") for a, b in zip(want, got))
                    ok += 1
                    delivered.add(j)
                    cursor = j + 1
                break
    missing_bits = sum(len(sent_tbs[j][1]) for j in range(len(sent_tbs)) if j not in delivered) * 8
    return dict(bits_sent=bits_sent, bit_errors=errors + missing_bits, blocks_sent=len(sent_tbs), blocks_ok=ok,
                ber=(errors + missing_bits) / bits_sent if bits_sent else None,
                bler=1.0 - ok / len(sent_tbs) if sent_tbs else None)
