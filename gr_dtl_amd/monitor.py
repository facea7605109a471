"""Monitor counters of the receive path, in the shape of the reference's telemetry messages
(lib/dtl/proto/monitor_ofdm.proto:3-22): one `monitor_eq_msg` per decoded frame
(ofdm_adaptive_frame_equalizer_vcvc_impl.cc:196-216) and one `monitor_dec_msg` per transport block
(ofdm_adaptive_fec_decoder_impl.cc:182-195).  The reference publishes them on message ports as protobuf
blobs; here they are plain dicts built from the record arrays of an `rx_result`, with the running counters
(CRC ok / fail, lost frames) carried across work() / feed() calls by a `monitor_probe`."""
from .ofdm_adaptive_config import get_bits_per_symbol


class monitor_probe:
    def __init__(self):
        self.crc_ok_count = 0
        self.crc_fail_count = 0
        self.frames_count = 0      # d_frames_count: decoded + lost
        self.lost_frames = 0       # d_lost_frames

    def eq_msgs(self, rx_result):
        """monitor_eq_msg per payload frame.  `lost_frames` in the frame records is cumulative over the stream."""
        out = []
        for f in rx_result.frames:
            if f["payload_index"] < 0:
                continue
            lost_now = int(f["lost_frames"]) - self.lost_frames if int(f["lost_frames"]) >= self.lost_frames \
                else int(f["lost_frames"])            # a new stream restarts the counter
            self.lost_frames += lost_now
            self.frames_count += lost_now + 1
            out.append(dict(constellation_key=int(f["feedback_cnst"]), fec_key=int(f["feedback_fec"]),
                            estimated_snr_tag_key=float(f["snr_db"]), noise_tag_key=float(f["noise"]),
                            lost_frames_rate=100.0 * self.lost_frames / self.frames_count))
        return out

    def dec_msgs(self, rx_result, code_n=0, code_k=0):
        """monitor_dec_msg per transport block (FEC chain) or per packed frame (no-FEC chain)."""
        bps_by_tb = {}
        payload_by_tb = {}
        for f in rx_result.frames:
            if f["payload_index"] >= 0:
                bps_by_tb.setdefault(int(f["tb_no"]), get_bits_per_symbol(int(f["cnst"])))
                payload_by_tb[int(f["tb_no"])] = int(f["payload_len"])
        out = []
        for t in rx_result.tbs:
            if t["crc_ok"]:
                self.crc_ok_count += 1
            else:
                self.crc_fail_count += 1
            total = self.crc_ok_count + self.crc_fail_count
            n_cw = 0
            if code_n:
                frame_len = 8 * payload_by_tb.get(int(t["tb_no"]), 0)
                n_cw = 1 + frame_len // code_n if frame_len > code_n else 1
            out.append(dict(tb_no=int(t["tb_no"]), tb_payload=int(t["tb_payload"]), tb_code_k=int(code_k),
                            tb_code_n=int(code_n), tb_codewords=n_cw,
                            frame_payload=payload_by_tb.get(int(t["tb_no"]), 0),
                            bps=bps_by_tb.get(int(t["tb_no"]), 0),
                            crc_ok_count=self.crc_ok_count, crc_fail_count=self.crc_fail_count,
                            tber=100.0 * self.crc_fail_count / total, avg_it=float(t["avg_it"])))
        return out


# ---------------------------------------------------------------------------------------------------------
# Wire form.  The reference wraps each message in a `monitor_proto_msg` envelope (lib/testbed/monitor.proto:
# time = 1, proto_id = 2, optional nmsgs = 3, optional sent_counter = 4, payload = 5 as google.protobuf.Any)
# built by monitor_proto<M, id>::build (include/gnuradio/testbed/monitor_proto.h:99-110) with the ids of
# lib/dtl/ofdm_adaptive_monitor.h:20-21; values are cast to the field's C++ type there (`tber` is an int32
# field fed with a double, so it truncates).  proto3 rules: scalar fields equal to zero are not emitted,
# int32 / int64 are two's-complement varints (negative values take ten bytes), doubles are little-endian
# fixed64.  Written out by hand: nothing on the product side needs a protobuf runtime.
import struct
import time as _time

FEC_DEC_MSG = 0
EQ_MSG = 1
TAG_PROTO = 0x5C       # include/gnuradio/testbed/monitor_parser.h:24

_DEC_FIELDS = (("tb_no", 1, "i"), ("tb_payload", 2, "i"), ("tb_code_k", 3, "i"), ("tb_code_n", 4, "i"),
               ("tb_codewords", 5, "i"), ("frame_payload", 6, "i"), ("bps", 7, "i"), ("crc_ok_count", 8, "i"),
               ("crc_fail_count", 9, "i"), ("tber", 10, "i"), ("avg_it", 11, "d"))
_EQ_FIELDS = (("constellation_key", 1, "i"), ("fec_key", 2, "i"), ("estimated_snr_tag_key", 3, "d"),
              ("noise_tag_key", 4, "d"), ("lost_frames_rate", 5, "d"))


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_int(no, v):
    v = int(v)
    return b"" if v == 0 else _varint(no << 3) + _varint(v)


def _field_double(no, v):
    bits = struct.pack("<d", float(v))
    return b"" if bits == b"\0" * 8 else _varint((no << 3) | 1) + bits      # -0.0 is emitted, as protobuf does


def _field_bytes(no, b):
    return _varint((no << 3) | 2) + _varint(len(b)) + b


def _payload(msg, fields):
    out = b""
    for name, no, kind in fields:
        v = msg.get(name, 0)
        if kind == "i":
            iv = int(v)                                   # C++ (int32_t)V: truncation toward zero
            iv = ((iv + (1 << 31)) % (1 << 32)) - (1 << 31)
            out += _field_int(no, iv)
        else:
            out += _field_double(no, v)
    return out


def encode_msg(msg, proto_id, time_ms=None, nmsgs=None, sent_counter=None):
    """Serialised `monitor_proto_msg` carrying one monitor_dec_msg (proto_id FEC_DEC_MSG) or monitor_eq_msg
    (EQ_MSG) dict from monitor_probe; `time` is the wall clock in milliseconds (monitor_msg.cc:18-22).  `nmsgs` / `sent_counter` are what monitor_probe_impl.cc:76-77 adds
    before the blob goes on the wire (explicit-presence fields: written even when zero)."""
    fields, type_name = (_DEC_FIELDS, "monitor_dec_msg") if proto_id == FEC_DEC_MSG else (_EQ_FIELDS, "monitor_eq_msg")
    any_msg = _field_bytes(1, ("type.googleapis.com/" + type_name).encode())
    body = _payload(msg, fields)
    if body:
        any_msg += _field_bytes(2, body)
    out = _field_int(1, _time.time_ns() // 1000000 if time_ms is None else time_ms) + _field_int(2, proto_id)
    if nmsgs is not None:
        out += _varint(3 << 3) + _varint(int(nmsgs))
    if sent_counter is not None:
        out += _varint(4 << 3) + _varint(int(sent_counter))
    return out + _field_bytes(5, any_msg)


def encode_probe_datagram(msg, proto_id, **kw):
    """The ZMQ datagram of monitor_probe_impl.cc:78-82: one tag byte (TAG_PROTO) + the envelope."""
    return bytes([TAG_PROTO]) + encode_msg(msg, proto_id, **kw)
