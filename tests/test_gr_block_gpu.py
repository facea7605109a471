"""The GNU Radio block of the drop-in -- shim/lib/ofdm_adaptive_rx_hip_impl.cc, the source a gr-dtl maintainer would
compile against GNU Radio 3.10 -- EXECUTED: compiled against the functional stand-in tests/cpp/mock_gnuradio/ (gr::block
bookkeeping, pmt, tags, message ports) and driven by tests/cpp/test_gr_block.cpp the way a scheduler thread drives
general_work: caller-owned buffers with guard words, ragged ninput_items (from nothing to several frames), small
noutput_items, consume / produce accounting.  Interface under test: the seven ports + message ports of the reference's hier
block (python/dtl/ofdm_adaptive_rx.py:42-59, grc/dtl_ofdm_adaptive_rx.block.yml:22-53), the length tags of the FEC decoder
(lib/dtl/ofdm_adaptive_fec_decoder_impl.cc:161-181) and the tag set of the frame equaliser
(lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:218-251; qa_ofdm_adaptive_frame_equalizer_vcvc.py:155-161).
Everything the block hands to the runtime is compared with the oracle on the same stream."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import CODES, ROOT
from helpers import make_pair, make_stream, oracle_feedback

MOCK = os.path.join(ROOT, "tests", "cpp", "mock_gnuradio")
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")


def build_block_harness():
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "test_gr_block")
    lib_dir = os.path.join(ROOT, "gr_dtl_amd")
    cmd = ["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-x", "c++",
           "-D__HIP_PLATFORM_AMD__", "-I", MOCK, "-I", os.path.join(ROOT, "shim", "include"), "-I", os.path.join(ROOT, "shim", "lib"),
           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "cpp", "test_gr_block.cpp"), os.path.join(ROOT, "shim", "lib", "ofdm_adaptive_rx_hip_impl.cc"),
           "-o", out, "-L", lib_dir, "-ldtl_hip", "-Wl,-rpath," + lib_dir, "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stderr[-4000:]
    return out


def write_cfg(path, pcfg, rx):
    """the flattened configuration as "key values" lines (what make_cfg_struct puts into dtl_rx_cfg)"""
    lines = ["fft_len %d" % pcfg.fft_len, "cp_len %d" % pcfg.cp_len, "frame_length %d" % pcfg.frame_length,
             "sync_threshold %r" % float(pcfg.sync_threshold), "use_sync_correct %d" % int(pcfg.use_sync_correct),
             "initial_mcs %d" % int(pcfg.initial_mcs_id)]
    for s in pcfg.occupied_carriers:
        lines.append("occ " + " ".join(str(int(c)) for c in s))
    for s in pcfg.pilot_carriers:
        lines.append("pilot_set " + " ".join(str(int(c)) for c in s))
    for row in pcfg.pilot_symbols:
        lines.append("pilot_sym " + " ".join("%r %r" % (float(np.real(v)), float(np.imag(v))) for v in row))
    for name, arr in (("sync_word1", pcfg.sync_word1), ("sync_word2", pcfg.sync_word2)):
        z = np.asarray(arr, dtype=np.complex64)
        lines.append(name + " " + " ".join("%r %r" % (float(v.real), float(v.imag)) for v in z))
    for _, alist in pcfg.fec_codes:
        lines.append("alist " + os.path.join(CODES, alist))
    lines.append("mcs_snr " + " ".join(repr(float(m[0])) for m in rx.mcs))
    lines.append("mcs_cnst " + " ".join(str(int(m[1][0])) for m in rx.mcs))
    lines.append("mcs_fec " + " ".join(str(int(m[1][1])) for m in rx.mcs))
    open(path, "w").write("\n".join(lines) + "\n")


def _value(text):
    kind, _, v = text.partition(":")
    return {"long": int, "u64": int, "double": float, "bool": lambda s: s == "1", "sym": str}[kind](v)


def read_outputs(outdir, N):
    ports = {0: np.fromfile(os.path.join(outdir, "port0.bin"), np.uint8),
             1: np.fromfile(os.path.join(outdir, "port1.bin"), np.complex64),
             2: np.fromfile(os.path.join(outdir, "port2.bin"), np.uint8),
             3: np.fromfile(os.path.join(outdir, "port3.bin"), np.uint8),
             4: np.fromfile(os.path.join(outdir, "port4.bin"), np.complex64).reshape(-1, N),
             5: np.fromfile(os.path.join(outdir, "port5.bin"), np.complex64),
             6: np.fromfile(os.path.join(outdir, "port6.bin"), np.float32)}
    tags = {p: [] for p in range(7)}
    for line in open(os.path.join(outdir, "tags.txt")):
        port, off, key, val = line.split()
        tags[int(port)].append((int(off), key, _value(val)))
    msgs = []
    for line in open(os.path.join(outdir, "msgs.txt")):
        w = line.split()
        if w[1] == "dict":
            msgs.append((w[0], {kv.split("=", 1)[0]: _value(kv.split("=", 1)[1]) for kv in w[2:]}))
        else:
            msgs.append((w[0], _value(w[1])))
    return ports, tags, msgs


HEADER_KEYS_FEC = ["fec_tb_key", "fec_feedback_key", "fec_offset_key", "fec_key", "fec_tb_payload_key"]
HEADER_KEYS = ["payload_length_key", "packet_length", "frame_no", "constellation_tag_key", "feedback_constellation_key",
               "ofdm_sync_carr_offset"]


def check_against_oracle(oracle, pcfg, ocfg, code, rx, x, ports, tags, msgs, propagate):
    ref = oracle.rx_stream(ocfg, code, x, feedback=oracle_feedback(oracle, pcfg, rx),
                           want=("frames", "trig", "angle", "eq_hard", "eq_soft"))
    L, N = pcfg.frame_length, pcfg.fft_len
    has_fec = code is not None
    # ---- port 0 + its length tags
    np.testing.assert_array_equal(ports[0], ref["user_bytes"])
    delivered = [t for t in ref["tbs"] if (t["crc_ok"] if has_fec else True) and t["user_bits"] >= 8]
    assert len(delivered) >= 5
    want_tags, at = [], 0
    for t in delivered:
        want_tags.append((at, "packet_length", t["user_bits"] // 8))
        at += t["user_bits"] // 8
    assert tags[0] == want_tags
    assert at == ports[0].size
    assert ports[1].size == 0
    # ---- ports 2, 3, 6: one item per input sample once the stream has been flushed
    assert ports[2].size == x.size and ports[3].size == x.size and ports[6].size == x.size
    np.testing.assert_array_equal(ports[2], ref["trig_fixed"])
    np.testing.assert_array_equal(ports[3], ref["trig_raw"])
    np.testing.assert_array_equal(ports[6].view(np.uint32), np.asarray(ref["angle"], np.float32).view(np.uint32))   # the held arg(P): identical bits
    assert len(np.unique(ref["angle"])) > 5
    # ---- ports 4, 5: equaliser outputs of the frames whose header was accepted
    ok = [i for i, f in enumerate(ref["frames"]) if f["header_ok"]]
    want_hard = np.concatenate([ref["eq_hard"][i].reshape(L, N) for i in ok])
    want_soft = np.concatenate([ref["eq_soft"][i].reshape(-1) for i in ok])
    assert ports[4].shape == want_hard.shape
    np.testing.assert_array_equal(ports[4], want_hard)
    assert ports[5].shape == want_soft.shape
    assert (ports[5].view(np.uint32) == want_soft.view(np.uint32)).all()
    # ---- tags of the frame equaliser on both outputs, frame by frame
    keys = (HEADER_KEYS_FEC if has_fec else []) + HEADER_KEYS + ["noise_tag_key"] + \
        (["estimated_snr_tag_key", "feedback_constellation_key", "fec_feedback_key"] if propagate else []) + ["frame_length"]
    for port, per_frame in ((4, L), (5, L * N)):
        got = tags[port]
        pos = 0
        for j, i in enumerate(ok):
            f = ref["frames"][i]
            if not f["hdr"]["payload_len"]:
                continue                                              # empty frame: the block's work() returns 0 items
            mine = got[pos:pos + len(keys)]
            pos += len(keys)
            assert [t[1] for t in mine] == keys, (port, j)
            assert all(t[0] == j * per_frame for t in mine), (port, j, mine[0][0])
            vals = [t[2] for t in mine]
            h = f["hdr"]
            want = ([h["tb_no"], h["fec_fb"], h["tb_offset"], h["fec"], h["tb_payload"]] if has_fec else []) + \
                [h["payload_len"], f["n_payload_syms"], h["frame_no"], h["cnst"], h["fb_cnst"], f["carr_offset"]]
            assert vals[:len(want)] == want, (port, j)
            rest = vals[len(want):]
            assert abs(rest[0] - f["noise_lin"]) <= 1e-5 * abs(f["noise_lin"])
            if propagate:
                assert abs(rest[1] - f["snr_db"]) <= 1e-6 * abs(f["snr_db"]) + 1e-6
                assert rest[2:4] == [f["fb_cnst"], f["fb_fec"]]
            assert rest[-1] == L
        assert pos == len(got), (port, pos, len(got))
    if propagate:   # qa_ofdm_adaptive_frame_equalizer_vcvc.py:155-161 feeds constellation + payload length (+ taps) and counts 7
        # tags: what the block ADDS to those is the noise estimate, the SNR estimate and the two feedback keys
        fed = {"constellation_tag_key", "payload_length_key"}
        added = keys[len(HEADER_KEYS_FEC if has_fec else []) + len(HEADER_KEYS):-1]   # behind the header's keys, before the length tag
        assert sorted(fed | set(added)) == sorted(["constellation_tag_key", "payload_length_key", "noise_tag_key",
                                                    "estimated_snr_tag_key", "feedback_constellation_key", "fec_feedback_key"])
    # ---- message ports
    hdr_msgs = [m for p, m in msgs if p == "header"]
    fb_msgs = [m for p, m in msgs if p == "feedback"]
    mon = [m for p, m in msgs if p == "monitor"]
    assert len(hdr_msgs) == ref["n_frames"]
    assert [m is not False for m in hdr_msgs] == [bool(f["header_ok"]) for f in ref["frames"]]
    assert [m["frame_no"] for m in hdr_msgs if m is not False] == [ref["frames"][i]["hdr"]["frame_no"] for i in ok]
    assert [(m["feedback_constellation_key"], m["fec_feedback_key"]) for m in fb_msgs] == \
        [(ref["frames"][i]["fb_cnst"], ref["frames"][i]["fb_fec"]) for i in ok]
    eq_mon = [m for m in mon if "estimated_snr_tag_key" in m]
    dec_mon = [m for m in mon if "tb_no" in m]
    assert len(eq_mon) == len(ok) and len(dec_mon) == len(ref["tbs"])
    for m, i in zip(eq_mon, ok):
        f = ref["frames"][i]
        assert abs(m["estimated_snr_tag_key"] - f["snr_db"]) <= 1e-6 * abs(f["snr_db"]) + 1e-6
        assert m["constellation_key"] == f["fb_cnst"] and m["fec_key"] == f["fb_fec"]
    assert [(m["tb_no"], m["crc_ok"], m["avg_it"]) for m in dec_mon] == [(t["tb_no"], bool(t["crc_ok"]), t["avg_it"]) for t in ref["tbs"]]
    return ref


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sync", "async", "async-propagate"])
def test_block_general_work_matches_oracle_on_a_c2_stream(oracle, dtl, cuda, tmp_path, mode):
    """BASELINE configs[1]'s link (64-subcarrier OFDM, QPSK, LDPC(300,152)) as a 150-frame stream with noise, fed to the
    block in ragged scheduler buffers with at most 900 items of output room per call."""
    binary = build_block_harness()
    pcfg, ocfg, code = make_pair(oracle, dtl)
    rx = dtl.ofdm_receiver(pcfg)
    user, hdrs, x = make_stream(oracle, ocfg, code, 2, 17000, 13.0, seed=404)
    x = np.concatenate([x, np.zeros(ocfg.frame_samples, np.complex64)])
    assert len(hdrs) >= 140
    write_cfg(str(tmp_path / "cfg.txt"), pcfg, rx)
    x.tofile(str(tmp_path / "iq.c64"))
    out = subprocess.run([binary, str(tmp_path / "cfg.txt"), str(tmp_path / "iq.c64"), str(tmp_path),
                          "0" if mode == "sync" else "1", "7", "900", "1" if mode.endswith("propagate") else "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:] + out.stdout[-1000:]
    m = re.match(r"OK calls=(\d+) short=(\d+) read=(\d+)", out.stdout)
    assert m and int(m.group(3)) == x.size and int(m.group(1)) > 100
    ports, tags, msgs = read_outputs(str(tmp_path), pcfg.fft_len)
    ref = check_against_oracle(oracle, pcfg, ocfg, code, rx, x, ports, tags, msgs, mode.endswith("propagate"))
    assert np.array_equal(ref["user_bytes"][:len(user)], user[:len(ref["user_bytes"])]) or sum(t["crc_ok"] for t in ref["tbs"]) >= 100


@pytest.mark.gpu
def test_block_without_fec_and_with_lost_frames(oracle, dtl, cuda, tmp_path):
    """The no-FEC tail (config #1's graph: hard demapper + frame_pack) through the block, on a stream with two frames blanked
    out (lost-frame accounting in the monitor messages) -- one length tag per frame, as ofdm_adaptive_frame_pack_bb puts them."""
    binary = build_block_harness()
    pcfg, ocfg, code = make_pair(oracle, dtl, fec=None)
    rx = dtl.ofdm_receiver(pcfg)
    user, hdrs, x = make_stream(oracle, ocfg, None, 3, 12000, 24.0, seed=77)
    fs = ocfg.frame_samples
    x = x.copy()
    x[100 + 9 * fs:100 + 10 * fs] = 0
    x[100 + 30 * fs:100 + 31 * fs] = 0
    x = np.concatenate([x, np.zeros(fs, np.complex64)])
    write_cfg(str(tmp_path / "cfg.txt"), pcfg, rx)
    x.tofile(str(tmp_path / "iq.c64"))
    out = subprocess.run([binary, str(tmp_path / "cfg.txt"), str(tmp_path / "iq.c64"), str(tmp_path), "1", "3", "400"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:] + out.stdout[-1000:]
    ports, tags, msgs = read_outputs(str(tmp_path), pcfg.fft_len)
    ref = check_against_oracle(oracle, pcfg, ocfg, None, rx, x, ports, tags, msgs, False)
    assert max(f["lost_frames"] for f in ref["frames"] if f["header_ok"]) >= 2
    rates = [m["lost_frames_rate"] for p, m in msgs if p == "monitor" and isinstance(m, dict) and "lost_frames_rate" in m]
    lost = [f["lost_frames"] for f in ref["frames"] if f["header_ok"]]
    want = [100.0 * l / (l + k + 1) for k, l in enumerate(lost)]     # frames_count = lost + frames seen (:124-138, :215)
    np.testing.assert_allclose(rates, want, rtol=1e-12)
