"""Frame dump format (lib/dtl/frame_file_store.cc:30-60) and the BER/FER report (tools/ber.py) -- host logic."""
import struct

import numpy as np


def test_record_format_and_long_count(dtl, tmp_path):
    fn = str(tmp_path / "tx.bin")
    with dtl.frame_file_store(fn) as st:
        st.store(3, 0, b"abc")
        st.store(2, 1, b"de")
        st.store(4, 4094, b"wxyz")      # jump of 4093 >= 5: skipped (1st skip)
        st.store(1, 2, b"f")
        st.store(1, 4095, b"g")         # an accepted frame refilled the skip budget: skipped (1st)
        st.store(1, 4095, b"g")         # 2nd
        st.store(1, 4095, b"g")         # 3rd
        st.store(1, 4095, b"g")         # skip budget exhausted: accepted, long count jumps
        # the 12-bit wrap 4095 -> 1 looks like a jump to the first test (|1 - 4095| >= 5): the reference skips
        # three frames before it re-synchronises its long count
        st.store(2, 1, b"hi")
        st.store(2, 2, b"jk")
        st.store(2, 3, b"lm")
        st.store(2, 4, b"no")           # accepted: increment (4095 - 4095) + 4 + 1 = 5
    blob = open(fn, "rb").read()
    recs, pos = [], 0
    while pos < len(blob):
        ln, n = struct.unpack_from("<iQ", blob, pos)
        recs.append((ln, n, blob[pos + 12:pos + 12 + ln]))
        pos += 12 + ln
    assert recs == [(3, 0, b"abc"), (2, 1, b"de"), (1, 2, b"f"), (1, 4095, b"g"), (2, 4100, b"no")]
    assert dtl.read_frames(fn) == [(n, p) for _, n, p in recs]


def test_ber_report_counts(dtl, tmp_path):
    rng = np.random.default_rng(3)
    frames = [bytes(rng.integers(0, 256, 20, dtype=np.uint8)) for _ in range(10)]
    txf, rxf = str(tmp_path / "tx.bin"), str(tmp_path / "rx.bin")
    with dtl.frame_file_store(txf) as tx:
        for i, p in enumerate(frames):
            tx.store(len(p), i, p)
    with dtl.frame_file_store(rxf) as rx:
        for i, p in enumerate(frames):
            if i in (0, 1, 6):
                continue                                   # RX starts late and loses frame 6
            q = bytearray(p)
            if i == 4:
                q[0] ^= 0x07                               # 3 bit errors
            rx.store(len(q), i, bytes(q))
    # RX's own long count starts at its first frame number: 2 (0 -> 2 is an increment of 2)
    r = dtl.ber_report(txf, rxf, tool_compat=False)
    assert r["frames_sent"] == 8 and r["frames_received"] == 7 and r["missed"] == 1
    assert r["crc_ok"] == 6 and r["crc_fail"] == 1
    assert r["bits_sent"] == 8 * 20 * 8 and r["bits_received"] == 7 * 20 * 8
    assert abs(r["ber_detected"] - 3 / (7 * 160)) < 1e-15
    assert abs(r["ber_overall"] - (3 + 160) / (8 * 160)) < 1e-15
    assert abs(r["fer"] - 2 / 8) < 1e-15


def test_monitor_probe_counters(dtl):
    from gr_dtl_amd.ofdm_receiver import _FRAME_DTYPE, _TB_DTYPE

    class R:
        pass
    r = R()
    r.frames = np.zeros(4, _FRAME_DTYPE)
    r.frames["payload_index"] = [0, -1, 1, 2]
    r.frames["lost_frames"] = [0, 0, 2, 2]
    r.frames["snr_db"] = [20.0, 0, 21.0, 22.0]
    r.frames["feedback_cnst"] = [2, 0, 2, 3]
    r.frames["cnst"] = [2, 0, 2, 2]
    r.frames["tb_no"] = [1, 0, 2, 3]
    r.frames["payload_len"] = [240, 0, 240, 240]
    r.tbs = np.zeros(3, _TB_DTYPE)
    r.tbs["tb_no"] = [1, 2, 3]
    r.tbs["crc_ok"] = [1, 0, 1]
    r.tbs["avg_it"] = [1, 15, 0]
    p = dtl.monitor_probe()
    eq = p.eq_msgs(r)
    assert [m["constellation_key"] for m in eq] == [2, 2, 3]
    assert [round(m["lost_frames_rate"], 6) for m in eq] == [0.0, round(100 * 2 / 4, 6), round(100 * 2 / 5, 6)]
    dec = p.dec_msgs(r, code_n=300, code_k=152)
    assert [(m["crc_ok_count"], m["crc_fail_count"]) for m in dec] == [(1, 0), (1, 1), (2, 1)]
    assert dec[1]["tber"] == 50.0 and dec[1]["avg_it"] == 15.0 and dec[0]["tb_codewords"] == 7 and dec[0]["bps"] == 2
    # counters continue over the next call (a continued stream)
    dec2 = p.dec_msgs(r)
    assert dec2[-1]["crc_ok_count"] == 4 and dec2[-1]["crc_fail_count"] == 2


# ---- pinned on the reference's own tool: tests/golden/ber_tool/ holds dumps written by frame_file_store and what
# /root/reference/tools/ber.py printed for them in the build container (tests/golden/make_ber_fixtures.py)
def _parse_ber_py(text):
    import re
    out = {}
    pats = {"mismatch_length": r"mismatch length=(\d+)", "missing_tx": r"missing txs=(\d+)", "matched": r"Matched frames:\s+(\d+)",
            "frames_sent": r"Sent: frames=(\d+)", "bits_sent": r"Sent: frames=\d+, bits=(\d+)",
            "frames_received": r"Received: frames=(\d+)", "bits_received": r"Received: frames=\d+, bits=(\d+)",
            "missed": r"missed=(\d+)", "crc_ok": r"crc_ok=(\d+)", "crc_fail": r"crc_fail=(\d+)",
            "ber_overall": r"BER \(overall\): (\S+)", "ber_detected": r"BER \(detected frames\): (\S+)", "fer": r"FER: (\S+)"}
    for k, p in pats.items():
        m = re.search(p, text)
        assert m, k
        out[k] = float(m.group(1)) if k in ("ber_overall", "ber_detected", "fer") else int(m.group(1))
    return out


def test_ber_report_equals_the_reference_tool(dtl, tmp_path):
    import importlib.util
    import os
    from conftest import GOLDEN
    spec = importlib.util.spec_from_file_location("make_ber_fixtures", os.path.join(GOLDEN, "make_ber_fixtures.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    cases = gen.scenarios()
    assert len(cases) >= 4
    for name, (tx, rx) in cases.items():
        base = os.path.join(GOLDEN, "ber_tool", name)
        # (a) the writer reproduces the committed dumps byte for byte
        for side, recs in (("tx", tx), ("rx", rx)):
            fn = str(tmp_path / (name + "." + side + ".bin"))
            gen.write_dump(fn, recs)
            assert open(fn, "rb").read() == open(base + "." + side + ".bin", "rb").read(), (name, side)
        # (b) ber_report gives the figures the reference's tool printed
        want = _parse_ber_py(open(base + ".ber_py.txt").read())
        got = dtl.ber_report(base + ".tx.bin", base + ".rx.bin")
        for k in ("frames_sent", "bits_sent", "frames_received", "bits_received", "missed", "crc_ok", "crc_fail",
                  "mismatch_length", "missing_tx"):
            assert got[k] == want[k], (name, k, got[k], want[k])
        for k in ("ber_overall", "ber_detected", "fer"):
            assert got[k] == want[k], (name, k, got[k], want[k])
