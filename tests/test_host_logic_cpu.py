"""Host-side (sequential, integer) logic of the product vs the oracle -- runs without a GPU."""
import ctypes as C
import json
import os

import numpy as np

from conftest import ALIST_100_23, ALIST_100_27, ALIST_300, GOLDEN
from helpers import make_pair

G = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))


def _fix_sync(dtl, frame_len, trig_pos, n):
    L = dtl._lib.lib()
    L.dtl_fix_sync.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    t = np.ascontiguousarray(trig_pos, dtype=np.int64)
    out = np.zeros(t.size * 3 + n // max(frame_len, 1) + 64, dtype=np.int64)
    n_out = C.c_int64(0)
    assert L.dtl_fix_sync(frame_len, t.ctypes.data, t.size, n, out.ctypes.data, out.size, C.byref(n_out)) == 0
    return out[:n_out.value]


def test_fix_sync_sparse_equals_dense_reference_form(oracle, dtl):
    rng = np.random.default_rng(5)
    FL = 1920
    for case in range(60):
        nfr = int(rng.integers(5, 80))
        pos = 172 + FL * np.arange(nfr)
        jit = rng.choice([0, 0, 0, 0, 1, -1, 2, -3, 5, -7, 12, 30], nfr) if case % 3 else np.zeros(nfr, int)
        pos = pos + jit
        keep = rng.random(nfr) > (0.15 if case % 2 else 0.0)      # missed detections
        pos = pos[keep]
        extra = rng.integers(0, FL * nfr, int(rng.integers(0, 4)))  # false alarms
        pos = np.unique(np.concatenate([pos, extra]))
        n = FL * nfr + 500
        pos = pos[(pos >= 0) & (pos < n)]
        dense = np.zeros(n, np.uint8)
        dense[pos] = 1
        ref, _ = oracle.fix_sync(FL, dense)
        got = _fix_sync(dtl, FL, pos, n)
        np.testing.assert_array_equal(got, np.nonzero(ref)[0], err_msg="case %d" % case)


def test_plateau_detector_equals_oracle(oracle, dtl):
    L = dtl._lib.lib()
    L.dtl_plateau_detect.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    pcfg, ocfg, code = make_pair(oracle, dtl)
    rng = np.random.default_rng(1)
    user = rng.integers(0, 256, 3000).astype(np.uint8)
    syms, hdrs = oracle.fec_frames_build(code, 1, 2, ocfg.frame_capacity, user)
    for nv, thr in ((0.0, 0.95), (0.3, 0.95), (1.5, 0.7), (4.0, 0.5)):
        ocfg2 = oracle.Config(sync_threshold=thr)
        x = oracle.awgn_stream(oracle.tx_frames(ocfg, syms, hdrs), nv, 3)
        trig, _, metric = oracle.sync_sc(ocfg2, x, want_metric=True)
        out = np.zeros(x.size // 16, np.int64)
        n_out = C.c_int64(0)
        assert L.dtl_plateau_detect(metric.ctypes.data, metric.size, C.c_float(thr), ocfg.cp_len, out.ctypes.data,
                                    out.size, C.byref(n_out)) == 0
        np.testing.assert_array_equal(out[:n_out.value], np.nonzero(trig)[0])


def test_feedback_decision_golden_and_random(oracle, dtl):
    L = dtl._lib.lib()
    L.dtl_feedback_sequence.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]

    def run(mcs, hyst, th, init, snr):
        th_ = np.array([m[0] for m in mcs], np.float64)
        cn = np.array([m[1][0] for m in mcs], np.int32)
        fe = np.array([m[1][1] for m in mcs], np.int32)
        snr = np.ascontiguousarray(snr, np.float64)
        oc, of = np.zeros(snr.size, np.int32), np.zeros(snr.size, np.int32)
        assert L.dtl_feedback_sequence(len(mcs), th_.ctypes.data, cn.ctypes.data, fe.ctypes.data, hyst, th, init,
                                       snr.ctypes.data, snr.size, oc.ctypes.data, of.ctypes.data) == 0
        return oc, of
    g = G["feedback_decision"]
    mcs = [(m[0], tuple(m[1])) for m in g["mcs"]]
    oc, _ = run(mcs, g["hysteresis"], g["decision_th"], g["initial_mcs"], g["snr_inputs"])
    assert oc.tolist() == g["expected_cnst"]
    rng = np.random.default_rng(2)
    mcs = [(-100000, (1, 1)), (11, (2, 1)), (16, (3, 1)), (21, (4, 1))]
    snr = np.cumsum(rng.normal(0, 1.5, 3000)) % 30
    fd = oracle.FeedbackDecision(1, 5, mcs, 1)
    ref = [fd.get_feedback(s) for s in snr]
    oc, of = run(mcs, 1.0, 5, 1, snr)
    assert list(zip(oc.tolist(), of.tolist())) == ref


def test_ldpc_code_tables_equal_oracle(oracle, dtl):
    rng = np.random.default_rng(3)
    for path in (ALIST_300, ALIST_100_23, ALIST_100_27):
        oc, pc = oracle.Ldpc(path), dtl.ldpc_dec(path)
        assert (pc.get_n(), pc.get_k()) == (oc.n, oc.k)
        np.testing.assert_array_equal(pc.permute(), oc.permute)
        data = rng.integers(0, 2, (20, oc.k)).astype(np.uint8)
        got = pc.encode_batch(data)
        for i in range(20):
            np.testing.assert_array_equal(got[i], oc.encode(data[i]))


def test_transmitter_equals_oracle_tx(oracle, dtl):
    from gr_dtl_amd.ofdm_transmitter import constellation_points, format_header, ofdm_transmitter
    rng = np.random.default_rng(4)
    for c in (1, 2, 3, 4):
        np.testing.assert_array_equal(constellation_points(c), oracle.constellation_points(c))
    for row, has_fec in ((G["packet_header_no_fec"], False), (G["packet_header_fec"], True)):
        for i, bits in enumerate(row["bits"]):
            f = dict(payload_len=row["packets_len"][i], frame_no=i, cnst=row["constellations"][i], fb_cnst=0)
            if has_fec:
                fe = row["fec"][i]
                f.update(tb_no=fe[0], tb_offset=fe[1], fec=fe[2], tb_payload=fe[3], fec_fb=0)
            assert format_header(f, has_fec, len(bits)).tolist() == bits
    for fft_len, cnst in ((64, 1), (64, 2), (64, 3), (64, 4), (256, 4)):
        pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=fft_len)
        tx = ofdm_transmitter(pcfg)
        user = rng.integers(0, 256, 3000 if fft_len == 64 else 9000).astype(np.uint8)
        s1, h1 = tx.build_fec_frames(user, cnst, "fec_1")
        s2, h2 = oracle.fec_frames_build(code, 1, cnst, ocfg.frame_capacity, user)
        np.testing.assert_array_equal(s1, s2)
        assert h1 == h2
        iq1, iq2 = tx.modulate(s1, h1), oracle.tx_frames(ocfg, s2, h2)
        assert np.abs(iq1 - iq2).max() <= 1e-6 * np.abs(iq2).max() + 2e-5
    pcfg, ocfg, _ = make_pair(oracle, dtl, fec=None)
    tx = ofdm_transmitter(pcfg)
    s1, h1 = tx.build_nofec_frames(user[:2500], 3)
    s2, h2 = oracle.nofec_frames_build(3, ocfg.frame_capacity, user[:2500])
    np.testing.assert_array_equal(s1, s2)


def test_config_mirror_defaults(dtl):
    """python/dtl/qa_ofdm_adaptive_config.py: defaults of the dataclass (fft_len 64 ...) and JSON overrides."""
    cfg = dtl.make_rx_config(None)
    assert (cfg.fft_len, cfg.cp_len, cfg.frame_length, cfg.sync_threshold) == (64, 16, 20, 0.95)
    assert len(cfg.occupied_carriers[0]) == 48 and cfg.pilot_carriers[0] == (-21, -7, 7, 21)
    assert len(cfg.pilot_symbols) == 127 and tuple(cfg.pilot_symbols[4]) == (-1, -1, -1, 1)
    assert len(cfg.sync_word1) == 64 and abs(cfg.sync_word1[32]) == 0
    assert np.count_nonzero(cfg.sync_word1) == 26 and np.count_nonzero(cfg.sync_word2) == 52
    cfg = dtl.make_rx_config({"mcs": [[-100000, ["bpsk", "fec_1"]], [11, ["qpsk", "fec_1"]]],
                              "initial_mcs_id": 1, "fec_codes": [["fec_1", "n_0300_k_0152_gap_03.alist"]],
                              "not_a_field": 3})
    assert cfg.mcs[1] == (11, (dtl.constellation_type_t.QPSK, "fec_1")) and cfg.initial_mcs_id == 1
    rx = dtl.ofdm_receiver(cfg)
    assert rx.fec == 1 and rx.header_len == 2 and rx.mcs[0][1] == (dtl.constellation_type_t.BPSK, 1)
    assert rx.frame_capacity == 960 and rx.frame_samples == 1920
    scaled = dtl.make_scaled_rx_config(256)
    assert len(scaled.occupied_carriers[0]) == 192 and len(scaled.pilot_carriers[0]) == 16


def test_tx_framing_state_machine_matches_oracle(oracle, dtl):
    """dtl_tx_plan (host only): the header fields of every frame the GPU transmitter would emit equal those of the
    oracle's restatement of ofdm_adaptive_fec_frame_bvb / ofdm_adaptive_frame_bb, for all constellations, odd sizes,
    frame lengths and numbering offsets."""
    import os
    from conftest import CODES
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    code = oracle.Ldpc(os.path.join(CODES, "n_0300_k_0152_gap_03.alist"))
    keys = ("payload_len", "frame_no", "cnst", "fb_cnst", "tb_no", "fec_fb", "tb_offset", "fec", "tb_payload")
    rng = np.random.default_rng(9)
    for fft_len, frame_length in ((64, 20), (64, 5), (256, 10)):
        cfg = dtl.ofdm_adaptive_rx_config(frame_length=frame_length) if fft_len == 64 else \
            dtl.make_scaled_rx_config(fft_len, frame_length=frame_length)
        cfg.fec_codes = (("fec_1", "n_0300_k_0152_gap_03.alist"),)
        tx = ofdm_transmitter(cfg)
        for cnst in (1, 2, 3, 4):
            for nbytes in (1, 117, 118, 119, 1000, int(rng.integers(2000, 9000))):
                user = rng.integers(0, 256, nbytes, dtype=np.uint8)
                _, ref = oracle.fec_frames_build(code, 1, cnst, tx.frame_capacity, user, first_frame_no=4000)
                got = tx.plan_device(nbytes, cnst, "fec_1", first_frame_no=4000)
                assert len(got) == len(ref), (fft_len, frame_length, cnst, nbytes)
                for k in keys:
                    assert list(got[k]) == [h[k] for h in ref], (fft_len, frame_length, cnst, nbytes, k)
    cfg = dtl.ofdm_adaptive_rx_config()
    tx = ofdm_transmitter(cfg)          # no FEC
    for cnst in (1, 2, 3, 4):
        user = rng.integers(0, 256, 5000, dtype=np.uint8)
        _, ref = oracle.nofec_frames_build(cnst, tx.frame_capacity, user, first_frame_no=3)
        got = tx.plan_device(5000, cnst, first_frame_no=3)
        assert [int(v) for v in got["payload_len"]] == [h["payload_len"] for h in ref]
        assert [int(v) for v in got["frame_no"]] == [h["frame_no"] for h in ref]


def test_product_header_formatter_against_reference_rows(dtl):
    """The product's host-side header formatter (the bits the GPU transmitter modulates) reproduces the six header
    bit rows of the reference's own QA test (tests/golden, qa_ofdm_adaptive_packet_header.py:73-260), CRC-16 included."""
    import ctypes as C
    import json
    import os
    from conftest import GOLDEN
    from gr_dtl_amd.ofdm_transmitter import TxFrameInfo
    G = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))
    L = dtl._lib.lib()
    L.dtl_header_format_host.argtypes = [C.POINTER(TxFrameInfo), C.c_int32, C.c_int32, C.c_void_p]
    for key, has_fec, nbits in (("packet_header_no_fec", 0, 48), ("packet_header_fec", 1, 96)):
        g = G[key]
        for i, bits in enumerate(g["bits"]):
            f = TxFrameInfo(payload_len=g["packets_len"][i], frame_no=i, cnst=g["constellations"][i], fb_cnst=0)
            if has_fec:
                f.tb_no, f.tb_offset, f.fec, f.tb_payload = g["fec"][i]
            out = np.zeros(nbits, np.uint8)
            assert L.dtl_header_format_host(C.byref(f), has_fec, nbits, out.ctypes.data) == 0
            assert out.tolist() == bits, (key, i)


def test_product_constellation_tables_against_reference_constants(oracle, dtl):
    """The tables the kernels slice / demap with: 8PSK[0] = e^{j pi/8}, BPSK[0] = -1 and the metric constant of the
    reference's QA test (qa_ofdm_adaptive_constellation_metric_vcvf.py:86); bit-identical to the oracle's tables."""
    import ctypes as C
    import json
    import os
    from conftest import GOLDEN
    G = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))
    L = dtl._lib.lib()
    L.dtl_constellation_points_host.argtypes = [C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
    pts = {}
    for cnst in (1, 2, 3, 4):
        buf = np.zeros(16, np.complex64)
        n = C.c_int32(0)
        assert L.dtl_constellation_points_host(cnst, buf.ctypes.data, C.byref(n)) == 0 and n.value == 1 << cnst
        pts[cnst] = buf[:n.value]
        np.testing.assert_array_equal(pts[cnst].view(np.uint32), oracle.constellation_points(cnst).view(np.uint32))
    p8, pb = pts[3], pts[1]
    dmin2 = min(abs(p8[i] - p8[j]) ** 2 for i in range(8) for j in range(i))
    assert abs(abs(p8[0] - pb[0]) ** 2 / dmin2 - G["constellation_metric"]["value"]) < 1e-5
    assert abs(p8[0] - np.exp(1j * np.pi / 8)) < 1e-6 and pb[0] == -1
    assert L.dtl_constellation_points_host(9, buf.ctypes.data, C.byref(n)) == dtl._lib.ERR_INVALID_ARG


def test_transport_block_bookkeeping_matches_oracle(oracle, dtl):
    """dtl_tb_gather without a descriptor table (host work only): which transport blocks a run of frame headers completes
    -- block numbers, payload sizes, codeword counts -- against the oracle's tb_decoder restatement fed with the same
    headers, over constellation changes, lost frames and frames cut out of the middle of a block."""
    from gr_dtl_amd.ofdm_receiver import _FRAME_DTYPE, TbSpan
    L = dtl._lib.lib()
    rng = np.random.default_rng(77)
    cap = 96
    compared = blocks = 0
    for case in range(40):
        alist = [ALIST_300, ALIST_100_23, ALIST_100_27][case % 3]
        code = oracle.Ldpc(alist)
        hdrs, cnsts = [], []
        frame_no = int(rng.integers(0, 4000))
        for seg in range(int(rng.integers(1, 4))):
            cnst = int(rng.integers(1, 5))
            if (max(1, 1 + (cap * cnst // 8 * 8) // code.n) * code.k) // 8 <= 4:
                cnst = 4
            user = rng.integers(0, 256, int(rng.integers(1, 40 * cap * cnst // 8)), dtype=np.uint8)
            _, hh = oracle.fec_frames_build(code, 1, cnst, cap, user, first_frame_no=frame_no)
            frame_no = (frame_no + len(hh)) & 0xFFF
            hdrs += hh
        keep = rng.random(len(hdrs)) > (0.0 if case % 4 == 0 else 0.12)           # lost frames
        hdrs = [h for h, k in zip(hdrs, keep) if k]
        if not hdrs:
            continue
        # product: frame records with consecutive soft-bit offsets
        fr = np.zeros(len(hdrs), dtype=_FRAME_DTYPE)
        off = 0
        tbd = oracle.TbDecoder(code, cap)
        want = []
        for i, h in enumerate(hdrs):
            bps = min(h["cnst"], 4)
            n_syms = min(-(-8 * h["payload_len"] // bps), cap)
            fr[i]["header_ok"] = 1
            for k in ("payload_len", "frame_no", "cnst", "fb_cnst", "tb_no", "fec_fb", "tb_offset", "fec", "tb_payload"):
                fr[i][k] = h[k]
            fr[i]["n_payload_syms"], fr[i]["payload_index"], fr[i]["llr_offset"] = n_syms, i, off
            off += n_syms * bps
            _, recs = tbd.frame(np.full(n_syms * bps, -5.0, np.float32), h["cnst"], h)   # soft bits of an all-zero word
            want += [(r["tb_no"], r["tb_payload"]) for r in recs]
        cn, ck = np.array([code.n], np.int32), np.array([code.k], np.int32)
        spans = (TbSpan * (2 * len(hdrs) + 4))()
        n_cw, n_sp = C.c_int64(0), C.c_int64(0)
        rc = L.dtl_tb_gather(cn.ctypes.data_as(C.c_void_p), ck.ctypes.data_as(C.c_void_p), 1, cap, 4, fr.ctypes.data_as(C.c_void_p),
                             C.c_int64(len(hdrs)), None, C.c_int64(1 << 40), C.byref(n_cw), spans, C.c_int64(len(spans)),
                             C.byref(n_sp), None)
        if rc == dtl._lib.ERR_UNSUPPORTED:     # a block of many pieces (frames lost inside it): the whole-chain call's job
            continue
        assert rc == 0, dtl._lib.last_error()
        got = [(spans[i].tb_no, spans[i].tb_payload_bits) for i in range(n_sp.value)]
        assert got == want, case
        assert n_cw.value == sum(spans[i].n_cw for i in range(n_sp.value))
        compared += 1
        blocks += len(got)
    assert compared >= 30 and blocks >= 300, (compared, blocks)


def test_fix_sync_fed_piecewise_equals_one_shot(oracle, dtl):
    """The receive path hands frame_detect_bb's walk the triggers range by range (under the sync kernels) and the
    buffer's length last (FixSyncRun): any split must give the one-shot result, over consecutive buffers too (the block's
    state -- in_sync, remainder, error counters -- carries over), and the one-shot result is the dense reference form's."""
    L = dtl._lib.lib()
    L.dtl_fix_sync_chunked.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]

    def chunked(FL, pos, ends, horizons):
        t = np.ascontiguousarray(pos, dtype=np.int64)
        e = np.ascontiguousarray(ends, dtype=np.int64)
        h = np.ascontiguousarray(sorted(horizons), dtype=np.int64)
        out = np.zeros(t.size * 3 + int(e[-1]) // max(FL, 1) + 64, dtype=np.int64)
        n_out = C.c_int64(0)
        assert L.dtl_fix_sync_chunked(FL, t.ctypes.data, t.size, e.ctypes.data, e.size, h.ctypes.data, h.size,
                                      out.ctypes.data, out.size, C.byref(n_out)) == 0
        return out[:n_out.value]
    rng = np.random.default_rng(11)
    for case in range(300):
        FL = int(rng.choice([1920, 1920, 640, 7680]))
        nfr = int(rng.integers(5, 120))
        pos = int(rng.integers(0, FL)) + FL * np.arange(nfr)
        if case % 4:
            pos = pos + rng.choice([0, 0, 0, 0, 0, 1, -1, 1, 2, -3, 5, -7, 9, 12, 30], nfr)
        if case % 3 == 0:       # a clock that drifts by a sample now and then (accumulated error -> corrections)
            pos = pos + np.cumsum(rng.random(nfr) < 0.15)
        pos = pos[rng.random(nfr) > (0.2 if case % 2 else 0.0)]                               # missed detections
        pos = np.unique(np.concatenate([pos, rng.integers(0, FL * nfr, int(rng.integers(0, 5)))]))  # false alarms
        n = FL * nfr + int(rng.integers(0, 2 * FL))
        pos = pos[(pos >= 0) & (pos < n)]
        # one buffer or several (cuts anywhere, also right behind a trigger and within the reach of a deadline)
        n_bufs = int(rng.choice([1, 1, 2, 3, 6]))
        ends = sorted(set(int(v) for v in rng.integers(1, n, n_bufs - 1))) + [n]
        one_shot = chunked(FL, pos, ends, [])
        for trial in range(3):
            nh = int(rng.choice([1, 3, 8, 40]))
            horizons = [int(v) for v in rng.integers(0, n + 1, nh)]
            if pos.size and trial == 1:
                horizons += [int(p) + int(d) for p, d in zip(rng.choice(pos, 5), rng.integers(-12, 13, 5))]
            got = chunked(FL, pos, ends, [h for h in horizons if 0 <= h <= n])
            np.testing.assert_array_equal(got, one_shot, err_msg="case %d trial %d" % (case, trial))
        if n_bufs == 1:
            dense = np.zeros(n, np.uint8)
            dense[pos] = 1
            ref, _ = oracle.fix_sync(FL, dense)
            np.testing.assert_array_equal(one_shot, np.nonzero(ref)[0], err_msg="case %d" % case)
