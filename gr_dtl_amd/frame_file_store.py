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


def _records(blob, start=0):
    """(long_count, payload_len, payload_offset) of every record from byte `start` on"""
    out, pos = [], start
    while pos + _REC.size <= len(blob):
        ln, n = _REC.unpack_from(blob, pos)
        pos += _REC.size
        out.append((n, ln, pos))
        pos += ln
    return out


def ber_report(tx_fname, rx_fname, tool_compat=True):
    """The figures tools/ber.py prints, as a dict: frames/bits sent and received, missed frames, frames
    without / with bit errors, BER over everything sent (missing frames count as all-wrong), BER over the
    received frames, FER.

    tool_compat=True (default) reproduces the reference's tool exactly, including where it reads the bytes it
    compares: tools/ber.py:104 adds the 12-byte record header to offsets that already point BEHIND the header
    (`headers()` yields f.tell() after the header, :15), so for every matched frame it compares bytes 12 .. 12+len
    behind the payload start in both files -- the payload without its first 12 bytes plus the 12 header bytes of the
    NEXT record (shorter at the end of a file).  A frame followed by different neighbours in the two dumps (a lost
    frame) is therefore counted as a frame in error.  Pinned on the tool's own output, tests/golden/ber_tool/.
    tool_compat=False compares payload with payload (what the tool means to do)."""
    with open(tx_fname, "rb") as f:
        tx_blob = f.read()
    with open(rx_fname, "rb") as f:
        rx_blob = f.read()
    empty = dict(frames_sent=0, frames_received=0, bits_sent=0, bits_received=0, missed=0, crc_ok=0, crc_fail=0,
                 ber_overall=None, ber_detected=None, fer=None, mismatch_length=0, missing_tx=0)
    # both dumps are read from the first RX record whose number the TX dump knows (:51-62)
    tx_all = _records(tx_blob)
    start_rx = start_tx = None
    for rx_n, _, rx_off in _records(rx_blob):
        hit = next((t_off for t_n, _, t_off in tx_all if t_n == rx_n), None)
        if hit is not None:
            start_rx, start_tx = rx_off - _REC.size, hit - _REC.size
            break
    if start_rx is None:
        return empty
    tx_headers, frames_sent, bits_sent = {}, 0, 0
    for n, ln, off in _records(tx_blob, start_tx):                      # :84-87 (a repeated number: the last wins)
        tx_headers[n] = (off, ln)
        bits_sent += ln * 8
        frames_sent += 1
    offsets, rx_headers = [], {}
    mismatch_length = missing_tx = 0
    for n, ln, off in _records(rx_blob, start_rx):                      # :90-102
        if n not in tx_headers:
            missing_tx += 1
            continue
        tx_off, tx_len = tx_headers[n]
        if tx_len != ln:
            mismatch_length += 1
            continue
        offsets.append((off, tx_off, ln))
        rx_headers[n] = (off, ln)
    missed = missing_bits = 0
    for n, ln, _ in _records(tx_blob, start_tx):                        # :104-107
        if n not in rx_headers:
            missed += 1
            missing_bits += ln * 8
    shift = _REC.size if tool_compat else 0
    errors = bits_received = frames_received = frame_errors = 0
    for rx_off, tx_off, ln in offsets:                                  # :113-127
        a = tx_blob[tx_off + shift:tx_off + shift + ln]
        b = rx_blob[rx_off + shift:rx_off + shift + ln]
        e = sum(bin(x ^ y).count("1") for x, y in zip(a, b))
        bits_received += ln * 8
        frames_received += 1
        errors += e
        frame_errors += e > 0
    return dict(frames_sent=frames_sent, frames_received=frames_received, bits_sent=bits_sent, bits_received=bits_received,
                missed=missed, crc_ok=frames_received - frame_errors, crc_fail=frame_errors,
                ber_overall=(errors + missing_bits) / bits_sent if bits_sent else None,
                ber_detected=errors / bits_received if bits_received else None,
                fer=(frame_errors + missed) / frames_sent if frames_sent else None,
                mismatch_length=mismatch_length, missing_tx=missing_tx)
