"""More streams no transmitter produced (next to tests/test_pathological_gpu.py), compared record by record: silence, a constant, pure noise, NaN / Inf samples, a signal with the preamble's
period everywhere (the timing metric sits at 1), bursts that make the metric cross its threshold thousands of times,
streams shorter than one symbol.  Whatever the reference's blocks do with them is what the oracle does; the HIP path
has to return the same triggers / frames / blocks -- or a status -- and the handle has to serve a regular stream
afterwards (both the single call and the call split into sample ranges)."""
import numpy as np
import pytest

from helpers import make_pair, make_stream, oracle_feedback


def _streams(ocfg, rng, regular):
    N, P = ocfg.fft_len, ocfg.fft_len + ocfg.cp_len
    n = 60 * P
    noise = (rng.normal(0, 1, n) + 1j * rng.normal(0, 1, n)).astype(np.complex64)
    half = (rng.normal(0, 1, N // 2) + 1j * rng.normal(0, 1, N // 2)).astype(np.complex64)
    periodic = np.tile(half, n // (N // 2))
    bursts = noise.copy()
    for s in range(0, n - 4 * P, 6 * P):                        # two symbols of period N/2, four of noise, ...
        bursts[s:s + 2 * P] = periodic[:2 * P]
    with_nan = regular.copy()
    with_nan[len(with_nan) // 3] = np.nan
    with_nan[len(with_nan) // 2] = np.complex64(complex(np.inf, -1.0))
    huge = (regular.astype(np.complex128) * 1e18).astype(np.complex64)
    tiny = (regular.astype(np.complex128) * 1e-18).astype(np.complex64)
    return {
        "silence": np.zeros(n, np.complex64),
        "constant": np.full(n, 0.7 - 0.2j, np.complex64),
        "noise": noise,
        "periodic": periodic,
        "bursts": bursts,
        "nan_inf": with_nan,
        "huge": huge,
        "tiny": tiny,
        "one_sample": regular[:1].copy(),
        "half_symbol": regular[100:100 + N // 2].copy(),
        "empty": regular[:0].copy(),
    }


def _records(res):
    return (np.asarray(res.raw_triggers), np.asarray(res.fixed_triggers),
            [(int(f["trigger"]), int(f["header_ok"]), int(f["payload_len"])) for f in res.frames],
            [(int(t["tb_no"]), int(t["crc_ok"]), int(t["user_bits"])) for t in res.tbs], res.payload.cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("split", [False, True])
def test_streams_no_transmitter_produced(oracle, dtl, cuda, monkeypatch, split):
    import torch
    from gr_dtl_amd import _lib
    if split:
        monkeypatch.setenv("DTL_RX_SPLIT_MIN", "3000")          # read when a receiver is created: the ranged sync stage
    pcfg, ocfg, code = make_pair(oracle, dtl)
    user, hdrs, regular = make_stream(oracle, ocfg, code, 2, 4000, 14.0, seed=5)
    rx = dtl.ofdm_receiver(pcfg)
    want_regular = _records(rx.work(torch.from_numpy(regular).to(cuda)))
    assert len(want_regular[3]) > 10 and want_regular[4].size > 0
    rng = np.random.default_rng(77)
    seen_status = 0
    for name, x in _streams(ocfg, rng, regular).items():
        ref = oracle.rx_stream(ocfg, code, x, feedback=oracle_feedback(oracle, pcfg, rx), want=("frames", "trig"))
        try:
            res = rx.work(torch.from_numpy(x).to(cuda))
        except _lib.DtlError as e:
            # a status is a legitimate answer where the reference would run out of buffer space or throw
            assert e.code in (_lib.ERR_CAPACITY, _lib.ERR_INVALID_ARG, _lib.ERR_TAGS_MISSING), (name, e)
            seen_status += 1
            res = None
        if res is not None:
            np.testing.assert_array_equal(res.raw_triggers, np.nonzero(ref["trig_raw"])[0], err_msg=name)
            np.testing.assert_array_equal(res.fixed_triggers, np.nonzero(ref["trig_fixed"])[0], err_msg=name)
            assert len(res.frames) == ref["n_frames"], name
            for f, rf in zip(res.frames, ref["frames"]):
                assert (f["trigger"], f["header_ok"]) == (rf["trigger"], rf["header_ok"]), name
            assert [(t["tb_no"], t["crc_ok"]) for t in res.tbs] == [(t["tb_no"], t["crc_ok"]) for t in ref["tbs"]], name
            np.testing.assert_array_equal(res.payload.cpu().numpy(), ref["user_bytes"], err_msg=name)
        # the handle serves a regular stream as if nothing had happened
        again = _records(rx.work(torch.from_numpy(regular).to(cuda)))
        for a, b in zip(again, want_regular):
            if isinstance(a, np.ndarray):
                np.testing.assert_array_equal(a, b, err_msg=name)
            else:
                assert a == b, name
    assert seen_status <= 2                                         # (most of them are ordinary streams without frames)


@pytest.mark.gpu
def test_odd_streams_fed_in_pieces_equal_one_call(oracle, dtl, cuda):
    """The same streams through feed() in ragged pieces: the sync kernel's tiles then lie differently relative to the bad
    samples, and the result must not notice (an output is lost only if its OWN window holds a term that is not finite)."""
    import torch
    from test_stream_gpu import _assert_same, _feed_all
    pcfg, ocfg, code = make_pair(oracle, dtl)
    user, hdrs, regular = make_stream(oracle, ocfg, code, 2, 6000, 14.0, seed=6)
    rng = np.random.default_rng(78)
    streams = _streams(ocfg, rng, regular)
    many = regular.copy()                                   # a bad sample every ~1.3 frames, at changing places
    many[np.arange(900, many.size, 2500)] = np.nan
    streams["many_nan"] = many
    for name in ("nan_inf", "many_nan", "bursts", "huge", "periodic"):
        x = streams[name]
        xt = torch.from_numpy(x).to(cuda)
        batch = dtl.ofdm_receiver(pcfg).work(xt)
        for cuts in ([x.size // 3, x.size // 2], list(range(4097, x.size, 4097)), list(range(777, x.size, 5000))):
            _assert_same(batch, _feed_all(dtl.ofdm_receiver(pcfg), xt, cuts))
    # and the many-NaN stream still delivers most of its blocks (every frame without a bad sample in its preamble, header
    # or payload decodes)
    res = dtl.ofdm_receiver(pcfg).work(torch.from_numpy(many).to(cuda))
    ref = oracle.rx_stream(ocfg, code, many, feedback=oracle_feedback(oracle, pcfg, dtl.ofdm_receiver(pcfg)), want=("frames", "trig"))
    np.testing.assert_array_equal(res.raw_triggers, np.nonzero(ref["trig_raw"])[0])
    assert [(t["tb_no"], t["crc_ok"]) for t in res.tbs] == [(t["tb_no"], t["crc_ok"]) for t in ref["tbs"]]
    np.testing.assert_array_equal(res.payload.cpu().numpy(), ref["user_bytes"])
    assert len(res.raw_triggers) >= 0.3 * len(hdrs)
