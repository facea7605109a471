"""Streaming work(): consecutive feed() calls that cut one IQ stream at arbitrary points must return the same
frames, transport blocks and payload bytes as one batch work() call over the whole stream (which the other
GPU tests pin against the oracle), with detector / oscillator / demux / transport-block / feedback state and
the samples of incomplete frames carried on the device between the calls."""
import numpy as np
import pytest

from helpers import make_pair, make_stream

FRAME_KEYS = ("trigger", "header_ok", "carr_offset", "payload_len", "frame_no", "cnst", "fb_cnst", "tb_no", "fec_fb",
              "tb_offset", "fec", "tb_payload", "n_payload_syms", "snr_db", "noise", "feedback_cnst", "feedback_fec",
              "lost_frames")
TB_KEYS = ("tb_no", "tb_payload", "crc_ok", "avg_it", "user_bits")


def _feed_all(rx, xt, cuts, want=()):
    """Feed xt in the chunks delimited by `cuts`; -> concatenated frames, tbs, payload, triggers, per-call results."""
    frames, tbs, payload, raw, fixed, angle, results = [], [], [], [], [], [], []
    bounds = [0] + list(cuts) + [xt.numel()]
    for i in range(len(bounds) - 1):
        r = rx.feed(xt[bounds[i]:bounds[i + 1]], want=want, final=(i == len(bounds) - 2))
        frames.append(r.frames)
        tbs.append(r.tbs)
        payload.append(r.payload.cpu().numpy())
        raw.append(r.raw_triggers)
        fixed.append(r.fixed_triggers)
        angle.append(r.sc_freq)
        results.append(r)
    return (np.concatenate(frames), np.concatenate(tbs), np.concatenate(payload), np.concatenate(raw),
            np.concatenate(fixed), np.concatenate(angle), results)


def _assert_same(batch, streamed):
    frames, tbs, payload, raw, fixed, angle, _ = streamed
    np.testing.assert_array_equal(raw, batch.raw_triggers)
    np.testing.assert_array_equal(fixed, batch.fixed_triggers)
    np.testing.assert_array_equal(angle.view(np.uint32), batch.sc_freq.view(np.uint32))
    assert len(frames) == len(batch.frames)
    for k in FRAME_KEYS:
        a, b = frames[k], batch.frames[k]
        if a.dtype.kind == "f":
            np.testing.assert_array_equal(a.view(np.uint64), b.view(np.uint64), err_msg=k)  # bit-identical
        else:
            np.testing.assert_array_equal(a, b, err_msg=k)
    assert len(tbs) == len(batch.tbs)
    for k in TB_KEYS:
        np.testing.assert_array_equal(tbs[k], batch.tbs[k], err_msg=k)
    np.testing.assert_array_equal(payload, batch.payload.cpu().numpy())


SCENARIOS = [
    # fft_len, frame_length, constellation, Es/N0, bytes, threshold, sync_correct, fec
    (64, 20, 2, 13.0, 12000, 0.95, False, True),
    (64, 20, 4, 24.0, 16000, 0.95, True, True),
    (64, 6, 3, 21.0, 5000, 0.95, True, True),
    (256, 10, 4, 24.0, 30000, 0.98, True, True),
    (64, 20, 2, 14.0, 6000, 0.95, True, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize("fft_len,frame_length,cnst,esn0,nbytes,thr,sync_correct,fec", SCENARIOS)
def test_chunked_feed_equals_batch(oracle, dtl, cuda, fft_len, frame_length, cnst, esn0, nbytes, thr, sync_correct, fec):
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=fft_len, frame_length=frame_length, sync_threshold=thr,
                                 use_sync_correct=sync_correct, fec="n_0300_k_0152_gap_03.alist" if fec else None)
    user, hdrs, x = make_stream(oracle, ocfg, code, cnst, nbytes, esn0, seed=77 + fft_len + cnst)
    xt = torch.from_numpy(x).to(cuda)
    rx = dtl.ofdm_receiver(pcfg)
    batch = rx.work(xt)
    assert batch.n_payload_frames >= 5 and len(batch.tbs) >= 3
    rng = np.random.default_rng(5)
    fs = rx.frame_samples
    cut_sets = [
        [xt.numel() // 2],                                                      # one cut
        list(range(fs // 3, xt.numel(), fs // 3)),                              # three calls per frame
        sorted(rng.choice(np.arange(1, xt.numel()), size=12, replace=False)),   # random cuts
        list(range(1000, xt.numel(), 5 * fs + 17)),                             # several frames per call
    ]
    compared = 0
    for cuts in cut_sets:
        rx2 = dtl.ofdm_receiver(pcfg)
        streamed = _feed_all(rx2, xt, cuts)
        if sync_correct:
            # frame_detect_bb looks ahead to the end of the buffer it is given (as in the reference): only
            # chunkings that leave its corrections alone are comparable
            if not np.array_equal(streamed[4], batch.fixed_triggers):
                continue
        _assert_same(batch, streamed)
        compared += 1
        assert rx2.stream_position == 0  # final=True reset the stream
    assert compared >= (2 if sync_correct else len(cut_sets))
    # the receiver object is reusable after a final call, and batch calls do not disturb anything
    again = rx.work(xt)
    np.testing.assert_array_equal(again.payload.cpu().numpy(), batch.payload.cpu().numpy())


@pytest.mark.gpu
def test_stream_without_sync_correct_is_cut_invariant(oracle, dtl, cuda):
    """use_sync_correct=False: no stage looks past its own samples, so EVERY chunking is bit-identical to the
    batch -- including tiny chunks, empty calls and a zero-length final flush."""
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=64, frame_length=20, use_sync_correct=False)
    user, hdrs, x = make_stream(oracle, ocfg, code, 4, 9000, 25.0, seed=4242)
    xt = torch.from_numpy(x).to(cuda)
    rx = dtl.ofdm_receiver(pcfg)
    batch = rx.work(xt)
    assert batch.n_payload_frames >= 5
    for step in (257, 1024, 4099):
        cuts = list(range(step, xt.numel(), step))
        _assert_same(batch, _feed_all(dtl.ofdm_receiver(pcfg), xt, cuts))
    # empty calls in the middle and an empty final call
    rx3 = dtl.ofdm_receiver(pcfg)
    half = xt.numel() // 2
    parts = [rx3.feed(xt[:half]), rx3.feed(xt[:0]), rx3.feed(xt[half:]), rx3.feed(xt[:0], final=True)]
    assert rx3.stream_position == 0
    payload = np.concatenate([p.payload.cpu().numpy() for p in parts])
    np.testing.assert_array_equal(payload, batch.payload.cpu().numpy())
    assert sum(len(p.frames) for p in parts) == len(batch.frames)
    # stream_position advances by what was fed; reset() drops the state
    rx4 = dtl.ofdm_receiver(pcfg)
    rx4.feed(xt[:5000])
    assert rx4.stream_position == 5000
    rx4.reset()
    assert rx4.stream_position == 0
    _assert_same(batch, _feed_all(rx4, xt, [12345]))


@pytest.mark.gpu
def test_stream_outputs_per_call(oracle, dtl, cuda):
    """Optional per-call outputs (equalised symbols, soft bits) of the streamed frames equal the batch's."""
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=64, frame_length=20, use_sync_correct=False)
    user, hdrs, x = make_stream(oracle, ocfg, code, 2, 6000, 15.0, seed=99)
    xt = torch.from_numpy(x).to(cuda)
    rx = dtl.ofdm_receiver(pcfg)
    want = ("eq_soft", "soft_syms", "llr", "metric")
    batch = rx.work(xt, want=want)
    cuts = list(range(3000, xt.numel(), 3000))
    streamed = _feed_all(dtl.ofdm_receiver(pcfg), xt, cuts, want=want)
    _assert_same(batch, streamed)
    eq = torch.cat([r.eq_soft for r in streamed[6] if r.n_payload_frames])
    assert torch.equal(eq.view(torch.float32), batch.eq_soft.view(torch.float32))
    metric = torch.cat([r.metric for r in streamed[6]])
    assert torch.equal(metric.view(torch.int32), batch.metric.view(torch.int32))
    # soft bits: frame by frame through each call's llr_offset
    b_llr = batch.llr.cpu().numpy()
    bi = 0
    b_frames = batch.frames[batch.frames["payload_index"] >= 0]
    for r in streamed[6]:
        llr = r.llr.cpu().numpy()
        for f in r.frames[r.frames["payload_index"] >= 0]:
            bf = b_frames[bi]
            bi += 1
            nb = min(int(f["n_payload_syms"]), rx.frame_capacity) * int(f["cnst"])
            np.testing.assert_array_equal(llr[f["llr_offset"]:f["llr_offset"] + nb].view(np.uint32),
                                          b_llr[bf["llr_offset"]:bf["llr_offset"] + nb].view(np.uint32))
    assert bi == len(b_frames)


@pytest.mark.gpu
def test_kernel_times_are_kept_per_call_and_read_afterwards(oracle, dtl, cuda):
    """dtl_rx_kernel_ms: the handle records HIP events around the phase-B and LDPC kernels of every call on that call's
    stream and keeps the last 32 calls; a caller that times several steps reads them behind its timed region."""
    import torch
    from gr_dtl_amd import _lib
    pcfg, ocfg, code = make_pair(oracle, dtl)
    user, hdrs, x = make_stream(oracle, ocfg, code, 2, 6000, 14.0, seed=4)
    rx = dtl.ofdm_receiver(pcfg)
    with pytest.raises(_lib.DtlError):
        rx.kernel_ms(0)                                  # no call yet
    xt = torch.from_numpy(x).to(cuda)
    for _ in range(3):
        res = rx.work(xt, copy_outputs=False)
    assert res.stage_ms[6] == 0.0 and res.stage_ms[7] == 0.0          # not read inside the call unless asked for
    for back in range(3):
        ldpc, payload = rx.kernel_ms(back)
        assert ldpc is not None and payload is not None and 0.0 < ldpc < 100.0 and 0.0 < payload < 100.0
    with pytest.raises(_lib.DtlError):
        rx.kernel_ms(3)                                  # only three calls so far
    res = rx.work(xt, want=("kernel_ms",), copy_outputs=False)
    assert res.stage_ms[6] > 0.0 and res.stage_ms[7] > 0.0
    empty = rx.work(torch.zeros(5000, dtype=torch.complex64, device=cuda), copy_outputs=False)   # nothing found: no kernels
    assert rx.kernel_ms(0) == (None, None)

