"""Frame dump files and the BER / FER report computed from a TX dump and an RX dump.

`frame_file_store` writes the record format of the reference's frame store
(lib/dtl/frame_file_store.cc:30-60): per stored frame `int32 payload_len | uint64 long_count | payload bytes`,
little endian, where long_count unwraps the 12-bit frame number of the packet header, and a frame whose number
jumps by 5 or more is ignored up to MAX_SKIP (3) times in a row.  The reference stores the user bytes of every
no-FEC frame on both sides (TX: ofdm_adaptive_frame_bb_impl.cc:286-288, RX: ofdm_adaptive_frame_pack_bb_impl.cc:110-114)
and tools/ber.py compares the two files; `ber_report` computes the same figures, so dumps written here can be fed
to the reference's tool and vice versa.
"""
import struct

MAX_HEADER_FRAME_NUMBER = 4095  # 12-bit frame number
MAX_SKIP = 3
_REC = struct.Struct("<iQ")


class frame_file_store:
    def __init__(self, fname=None):
        self._f = open(fname, "wb") if fname else None
        self.frame_last_number = 0
        self.frame_long_count = 0
        self.skip_count = MAX_SKIP

    def store(self, payload_len, count, payload):
        count = int(count)
        if abs(count - self.frame_last_number) >= 5 and self.skip_count:
            self.skip_count -= 1
            return
        if count >= self.frame_last_number:
            inc = count - self.frame_last_number
        else:
            inc = (MAX_HEADER_FRAME_NUMBER - self.frame_last_number) + count + 1
        if inc >= 5 and self.skip_count:
            self.skip_count -= 1
            return
        self.skip_count = MAX_SKIP
        self.frame_long_count += inc
        self.frame_last_number = count
        if self._f is not None:
            data = bytes(payload[:payload_len])
            self._f.write(_REC.pack(int(payload_len), self.frame_long_count))
            self._f.write(data)

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_frames(fname):
    """-> list of (long_count, payload bytes)"""
    out = []
    with open(fname, "rb") as f:
        blob = f.read()
    pos = 0
    while pos + _REC.size <= len(blob):
        ln, n = _REC.unpack_from(blob, pos)
        pos += _REC.size
        out.append((n, blob[pos:pos + ln]))
        pos += ln
    return out


def store_tx_frames(store, user_bytes, frame_user_bytes, first_frame_no=0):
    """TX side of the no-FEC chain: one record per frame with that frame's user bytes, numbered by the
    12-bit frame counter (ofdm_adaptive_frame_bb_impl.cc:286-288)."""
    data = bytes(user_bytes)
    for f, lo in enumerate(range(0, len(data), frame_user_bytes)):
        chunk = data[lo:lo + frame_user_bytes]
        store.store(len(chunk), (first_frame_no + f) & 0xFFF, chunk)


def store_rx_frames(store, rx_result):
    """RX side of the no-FEC chain: every frame the packer emitted (CRC ok or not), numbered by the frame
    number of its header (ofdm_adaptive_frame_pack_bb_impl.cc:110-114).  rx_result: ofdm_receiver.work()."""
    payload = rx_result.payload_bytes().tobytes() if hasattr(rx_result, "payload_bytes") else bytes(rx_result.payload)
    for t in rx_result.tbs:
        n = int(t["tb_payload"]) - 4
        if n < 0:
            continue
        off = int(t["out_offset"])
        store.store(n & 0xFFF, int(t["tb_no"]), payload[off:off + n])


def ber_report(tx_fname, rx_fname):
    """The figures tools/ber.py prints, as a dict: frames/bits sent and received, missed frames, frames
    without / with bit errors, BER over everything sent (missing frames count as all-wrong), BER over the
    received frames, FER."""
    tx = read_frames(tx_fname)
    rx = read_frames(rx_fname)
    tx_index = {n: i for i, (n, _) in enumerate(tx)}
    # both dumps are compared from the first RX frame that the TX dump knows
    start_rx = next((i for i, (n, _) in enumerate(rx) if n in tx_index), None)
    if start_rx is None:
        return dict(frames_sent=0, frames_received=0, bits_sent=0, bits_received=0, missed=0, crc_ok=0, crc_fail=0,
                    ber_overall=None, ber_detected=None, fer=None, mismatch_length=0, missing_tx=0)
    tx = tx[tx_index[rx[start_rx][0]]:]
    rx = rx[start_rx:]
    tx_map = {n: p for n, p in tx}
    frames_sent = len(tx)
    bits_sent = sum(len(p) for _, p in tx) * 8
    matched = {}
    mismatch_length = missing_tx = 0
    for n, p in rx:
        if n not in tx_map:
            missing_tx += 1
        elif len(tx_map[n]) != len(p):
            mismatch_length += 1
        else:
            matched[n] = p
    missed = [n for n, _ in tx if n not in matched]
    missing_bits = sum(len(tx_map[n]) for n in missed) * 8
    errors = bits_received = frame_errors = 0
    for n, p in matched.items():
        e = sum(bin(a ^ b).count("1") for a, b in zip(tx_map[n], p))
        errors += e
        bits_received += len(p) * 8
        frame_errors += e > 0
    return dict(frames_sent=frames_sent, frames_received=len(matched), bits_sent=bits_sent, bits_received=bits_received,
                missed=len(missed), crc_ok=len(matched) - frame_errors, crc_fail=frame_errors,
                ber_overall=(errors + missing_bits) / bits_sent if bits_sent else None,
                ber_detected=errors / bits_received if bits_received else None,
                fer=(frame_errors + len(missed)) / frames_sent if frames_sent else None,
                mismatch_length=mismatch_length, missing_tx=missing_tx)
