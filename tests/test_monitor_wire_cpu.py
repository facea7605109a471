"""Wire form of the monitor messages: the hand-written encoder of gr_dtl_amd/monitor.py against the protobuf
runtime (test side only), with the message schemas of lib/dtl/proto/monitor_ofdm.proto:3-22 and
lib/testbed/monitor.proto:5-15 declared here as descriptors."""
import struct

import numpy as np
import pytest

pb = pytest.importorskip("google.protobuf")
from google.protobuf import any_pb2, descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

from gr_dtl_amd import monitor  # noqa: E402

T = descriptor_pb2.FieldDescriptorProto


def _schema():
    f = descriptor_pb2.FileDescriptorProto(name="monitor_wire_test.proto", syntax="proto3",
                                           dependency=["google/protobuf/any.proto"])
    dec = f.message_type.add(name="monitor_dec_msg")
    for i, n in enumerate(["tb_no", "tb_payload", "tb_code_k", "tb_code_n", "tb_codewords", "frame_payload", "bps",
                           "crc_ok_count", "crc_fail_count", "tber"], 1):
        dec.field.add(name=n, number=i, type=T.TYPE_INT32, label=T.LABEL_OPTIONAL)
    dec.field.add(name="avg_it", number=11, type=T.TYPE_DOUBLE, label=T.LABEL_OPTIONAL)
    eq = f.message_type.add(name="monitor_eq_msg")
    eq.field.add(name="constellation_key", number=1, type=T.TYPE_INT32, label=T.LABEL_OPTIONAL)
    eq.field.add(name="fec_key", number=2, type=T.TYPE_INT32, label=T.LABEL_OPTIONAL)
    for i, n in enumerate(["estimated_snr_tag_key", "noise_tag_key", "lost_frames_rate"], 3):
        eq.field.add(name=n, number=i, type=T.TYPE_DOUBLE, label=T.LABEL_OPTIONAL)
    env = f.message_type.add(name="monitor_proto_msg")
    env.field.add(name="time", number=1, type=T.TYPE_INT64, label=T.LABEL_OPTIONAL)
    env.field.add(name="proto_id", number=2, type=T.TYPE_INT32, label=T.LABEL_OPTIONAL)
    env.oneof_decl.add(name="optional_nmsgs")
    env.oneof_decl.add(name="optional_sent_counter")
    env.field.add(name="nmsgs", number=3, type=T.TYPE_INT32, label=T.LABEL_OPTIONAL, oneof_index=0)
    env.field.add(name="sent_counter", number=4, type=T.TYPE_INT64, label=T.LABEL_OPTIONAL, oneof_index=1)
    env.field.add(name="payload", number=5, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".google.protobuf.Any")
    pool = descriptor_pool.DescriptorPool()
    pool.AddSerializedFile(any_pb2.DESCRIPTOR.serialized_pb)
    pool.AddSerializedFile(f.SerializeToString())
    get = message_factory.GetMessageClass
    return {n: get(pool.FindMessageTypeByName(n)) for n in ("monitor_dec_msg", "monitor_eq_msg", "monitor_proto_msg")}


def test_dec_message_round_trip_and_canonical_bytes():
    cls = _schema()
    msg = dict(tb_no=7, tb_payload=984, tb_code_k=152, tb_code_n=300, tb_codewords=52, frame_payload=1950, bps=4,
               crc_ok_count=300, crc_fail_count=0, tber=12.9, avg_it=1.25)
    blob = monitor.encode_msg(msg, monitor.FEC_DEC_MSG, time_ms=1696300000123, nmsgs=0, sent_counter=41)
    env = cls["monitor_proto_msg"].FromString(blob)
    assert env.time == 1696300000123 and env.proto_id == 0            # proto_id 0 is not on the wire: default
    assert env.HasField("nmsgs") and env.nmsgs == 0 and env.sent_counter == 41
    assert env.payload.type_url == "type.googleapis.com/monitor_dec_msg"
    dec = cls["monitor_dec_msg"].FromString(env.payload.value)
    for k, v in msg.items():
        assert getattr(dec, k) == (int(v) if k != "avg_it" else v), k   # tber: double cast to the int32 field
    # the runtime serialises the same message to the same bytes (field order, omitted zeros)
    ref = cls["monitor_proto_msg"](time=1696300000123, proto_id=0, nmsgs=0, sent_counter=41)
    ref.payload.type_url = env.payload.type_url
    ref.payload.value = cls["monitor_dec_msg"](**{k: (int(v) if k != "avg_it" else v) for k, v in msg.items()}).SerializeToString()
    assert ref.SerializeToString(deterministic=True) == blob


def test_eq_message_negative_and_zero_fields():
    cls = _schema()
    msg = dict(constellation_key=4, fec_key=-1, estimated_snr_tag_key=-3.5, noise_tag_key=0.0, lost_frames_rate=2.0)
    blob = monitor.encode_msg(msg, monitor.EQ_MSG, time_ms=5)
    env = cls["monitor_proto_msg"].FromString(blob)
    assert env.proto_id == 1 and not env.HasField("nmsgs") and not env.HasField("sent_counter")
    eq = cls["monitor_eq_msg"].FromString(env.payload.value)
    assert (eq.constellation_key, eq.fec_key, eq.estimated_snr_tag_key, eq.noise_tag_key, eq.lost_frames_rate) == (4, -1, -3.5, 0.0, 2.0)
    want = cls["monitor_eq_msg"](constellation_key=4, fec_key=-1, estimated_snr_tag_key=-3.5, lost_frames_rate=2.0)
    assert env.payload.value == want.SerializeToString(deterministic=True)
    dgram = monitor.encode_probe_datagram(msg, monitor.EQ_MSG, time_ms=5, nmsgs=3, sent_counter=9)
    assert dgram[0] == 0x5C and cls["monitor_proto_msg"].FromString(dgram[1:]).nmsgs == 3


def test_probe_messages_encode():
    """Messages built from result records go through the encoder (numpy scalars included)."""
    cls = _schema()
    frames = np.zeros(2, dtype=[("payload_index", "i4"), ("lost_frames", "i4"), ("feedback_cnst", "i4"), ("feedback_fec", "i4"),
                                ("snr_db", "f4"), ("noise", "f4"), ("tb_no", "i4"), ("cnst", "i4"), ("payload_len", "i4")])
    frames["payload_index"] = [0, 1]
    frames["lost_frames"] = [0, 3]
    frames["feedback_cnst"] = 4
    frames["snr_db"] = [21.5, 22.25]
    frames["noise"] = [0.01, 0.02]

    class R:
        pass
    r = R()
    r.frames = frames
    r.tbs = np.zeros(0, dtype=[("tb_no", "i4"), ("crc_ok", "i4"), ("tb_payload", "i4"), ("avg_it", "f4")])
    probe = monitor.monitor_probe()
    msgs = probe.eq_msgs(r)
    assert [m["lost_frames_rate"] for m in msgs] == [0.0, 100.0 * 3 / 5]
    for m in msgs:
        env = cls["monitor_proto_msg"].FromString(monitor.encode_msg(m, monitor.EQ_MSG))
        eq = cls["monitor_eq_msg"].FromString(env.payload.value)
        assert eq.estimated_snr_tag_key == m["estimated_snr_tag_key"] and eq.lost_frames_rate == m["lost_frames_rate"]
        assert struct.pack("<d", eq.noise_tag_key) == struct.pack("<d", m["noise_tag_key"])
