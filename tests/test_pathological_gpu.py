"""Inputs no transmitter produces: the receive path must neither fault nor hang, stay consistent with the oracle on
the integer outputs where the oracle is defined, and stay usable afterwards.  (A kernel fault on the GPU box can
take the node down, so every case is small.)"""
import numpy as np
import pytest

from helpers import make_pair, make_stream


def _cases(rng, frame):
    n = 30000
    t = np.arange(n)
    yield "zeros", np.zeros(n, np.complex64)
    yield "dc", np.full(n, 0.5 + 0.25j, np.complex64)                       # metric == 1 everywhere: one endless plateau
    yield "tone", np.exp(2j * np.pi * 0.03125 * t).astype(np.complex64)     # periodic with N/2: endless plateau too
    yield "huge", (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * np.float32(1e18)
    yield "tiny", (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * np.float32(1e-30)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    x[1000:1010] = np.nan
    x[5000] = np.inf
    yield "nan_inf", x
    yield "frame_repeated", np.tile(frame[:2000], 15).astype(np.complex64)  # preambles at a wrong spacing
    yield "one_sample", np.ones(1, np.complex64)
    yield "shorter_than_fft", np.ones(63, np.complex64)


@pytest.mark.gpu
@pytest.mark.parametrize("sync_correct", [True, False])
def test_pathological_streams(oracle, dtl, cuda, sync_correct):
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl, use_sync_correct=sync_correct)
    user, hdrs, good = make_stream(oracle, ocfg, code, 2, 3000, 16.0, seed=8)
    rx = dtl.ofdm_receiver(pcfg)
    want = rx.work(torch.from_numpy(good).to(cuda)).payload.cpu().numpy()
    assert want.size > 2000
    rng = np.random.default_rng(12)
    for name, x in _cases(rng, good[100:]):
        xt = torch.from_numpy(x).to(cuda)
        res = rx.work(xt)                       # must return (errors are allowed to be reported, not to crash)
        assert res.payload.numel() == res.n_payload_bytes, name
        # (round 4: also for "nan_inf" and "huge" -- a term that is not finite spoils the outputs whose own window holds it,
        # in the oracle and in the kernel alike; tests/test_odd_streams_gpu.py)
        ref = oracle.rx_stream(ocfg, code, x, want=("frames", "trig"))
        np.testing.assert_array_equal(res.raw_triggers, np.nonzero(ref["trig_raw"])[0], err_msg=name)
        assert len(res.frames) == ref["n_frames"], name
        # the same input through the streaming entry point in odd slices
        rx.reset()
        got = 0
        for lo in range(0, max(x.size, 1), 7001):
            r = rx.feed(xt[lo:lo + 7001], final=(lo + 7001 >= x.size))
            got += r.n_payload_bytes
        assert got == res.n_payload_bytes or sync_correct, name
        # ... and the receiver still decodes a good stream afterwards
        again = rx.work(torch.from_numpy(good).to(cuda)).payload.cpu().numpy()
        np.testing.assert_array_equal(again, want, err_msg=name)


@pytest.mark.gpu
def test_header_that_does_not_fit_is_refused(oracle, dtl, cuda):
    """40 data carriers: the 96 header bits (FEC) do not fit two header symbols -- the reference's formatter would cut the
    fields off and nothing would ever decode; the receiver says so instead."""
    import torch
    pcfg = dtl.ofdm_adaptive_rx_config(frame_length=4)
    pcfg.occupied_carriers = (tuple(pcfg.occupied_carriers[0])[4:-4],)
    pcfg.fec_codes = (("fec_1", "n_0300_k_0152_gap_03.alist"),)
    pcfg.mcs = ((-100000, (dtl.constellation_type_t.BPSK, "fec_1")),)
    rx = dtl.ofdm_receiver(pcfg)
    with pytest.raises(dtl._lib.DtlError) as e:
        rx.work(torch.zeros(5000, dtype=torch.complex64, device=cuda))
    assert e.value.code == dtl._lib.ERR_INVALID_ARG and "header" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("drop_db", [60, 90, 100])
def test_weak_frames_behind_a_strong_burst(oracle, dtl, cuda, drop_db):
    """The sync kernel treats a window whose energy is at or below 2^-38 (-114 dB) of its 4096-sample tile's energy as
    "no signal" (the rounding residue of the tile's running double sums; the reference's exact FIR sums have none).  A
    frame stream up to 100 dB below a burst that precedes it in the same tiles must still be found exactly as the oracle
    finds it, and -- whatever the level -- identically by one batch call and by a streamed run cut anywhere: tiles are
    aligned to the stream's sample 0, not to the call."""
    import torch
    from test_stream_gpu import _assert_same, _feed_all
    pcfg, ocfg, code = make_pair(oracle, dtl, sync_threshold=0.9)
    user, hdrs, x = make_stream(oracle, ocfg, code, 2, 2500, None, seed=21)      # noise-free frames
    rng = np.random.default_rng(3)
    burst = (rng.standard_normal(3000) + 1j * rng.standard_normal(3000)).astype(np.complex64) * np.float32(40.0)
    weak = (x * np.float32(40.0 * 10 ** (-drop_db / 20))).astype(np.complex64)
    y = np.concatenate([burst, weak]).astype(np.complex64)
    yt = torch.from_numpy(y).to(cuda)
    batch = dtl.ofdm_receiver(pcfg).work(yt)
    ref = oracle.rx_stream(ocfg, code, y, want=("frames", "trig"))
    np.testing.assert_array_equal(batch.raw_triggers, np.nonzero(ref["trig_raw"])[0])
    assert len(batch.frames) == ref["n_frames"] >= len(hdrs) - 2           # (the frame under the burst's tail may be lost)
    np.testing.assert_array_equal(batch.payload.cpu().numpy(), ref["user_bytes"])
    for cuts in ([2999], [3000, 3001, 7096], [4096, 8192], [1234, 5000, 9999, 20000]):
        streamed = _feed_all(dtl.ofdm_receiver(pcfg), yt, [c for c in cuts if c < y.size])
        if pcfg.use_sync_correct and not np.array_equal(streamed[4], batch.fixed_triggers):
            np.testing.assert_array_equal(streamed[3], batch.raw_triggers)  # (frame_detect_bb looks ahead to the buffer's end)
            continue
        _assert_same(batch, streamed)
