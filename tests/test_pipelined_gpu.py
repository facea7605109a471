"""The pipelined form of a call (sync stage in four sample ranges, arg(P) on a side stream, header kernels queued
for raw triggers before frame_detect_bb has seen the whole list) normally starts at 2^26 samples per call.  Here
the threshold is lowered (DTL_RX_SPLIT_MIN, read when a receiver is created) so that small streams take that path,
and the results are compared with the oracle and, record by record, with the unsplit path -- including streams
where frame_detect_bb moves / inserts triggers (jitter, lost preambles), a carrier-frequency offset, and
chunked feed() calls that carry triggers over."""
import numpy as np
import pytest

from helpers import make_pair, make_stream, oracle_feedback
from test_stream_gpu import FRAME_KEYS, TB_KEYS, _assert_same, _feed_all


@pytest.fixture()
def split_env(monkeypatch):
    monkeypatch.setenv("DTL_RX_SPLIT_MIN", "20000")


def _jittered(oracle, ocfg, code, cnst, n_bytes, esn0, seed, drop_every=0, jitter=0):
    """Frames with a few samples inserted between them (trigger jitter for frame_detect_bb) and some preambles blanked."""
    user, hdrs, x = make_stream(oracle, ocfg, code, cnst, n_bytes, esn0, seed)
    if not (drop_every or jitter):
        return user, x
    fs = (2 + 2 + ocfg.frame_length) * (ocfg.fft_len + ocfg.cp_len)
    rng = np.random.default_rng(seed + 7)
    parts = [x[:100]]
    for f in range((len(x) - 100) // fs):
        fr = x[100 + f * fs:100 + (f + 1) * fs].copy()
        if drop_every and f % drop_every == drop_every - 1:
            fr[:ocfg.fft_len + ocfg.cp_len] *= 0.05          # first sync word nearly gone: the detector misses it
        parts.append(fr)
        if jitter and f % 5 == 2:
            n_ins = int(rng.integers(1, jitter + 1))
            parts.append((rng.normal(0, 1e-3, n_ins) + 1j * rng.normal(0, 1e-3, n_ins)).astype(np.complex64))
    parts.append(x[100 + ((len(x) - 100) // fs) * fs:])
    return user, np.concatenate(parts)


@pytest.mark.gpu
@pytest.mark.parametrize("fft_len,cnst,esn0,n_bytes,drop,jitter,cfo", [
    (64, 2, 13.0, 20000, 0, 0, 0.0),
    (64, 4, 24.0, 24000, 7, 2, 0.0),
    (64, 3, 21.0, 16000, 0, 3, 0.3),
    (256, 4, 24.0, 40000, 5, 1, 0.0),
])
def test_pipelined_call_matches_oracle_and_unsplit(oracle, dtl, cuda, split_env, monkeypatch, fft_len, cnst, esn0, n_bytes,
                                                  drop, jitter, cfo):
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=fft_len, sync_threshold=0.95 if fft_len == 64 else 0.98)
    user, x = _jittered(oracle, ocfg, code, cnst, n_bytes, esn0, seed=31 + cnst, drop_every=drop, jitter=jitter)
    if cfo:
        x = (x * np.exp(2j * np.pi * cfo / fft_len * np.arange(len(x)))).astype(np.complex64)
    assert len(x) > 4 * 20000
    xt = torch.from_numpy(x).to(cuda)
    rx = dtl.ofdm_receiver(pcfg)                               # pipelined
    res = rx.work(xt)
    monkeypatch.setenv("DTL_RX_SPLIT_MIN", str(1 << 40))
    rx_plain = dtl.ofdm_receiver(pcfg)                         # one range, no early launches
    ref_gpu = rx_plain.work(xt)
    np.testing.assert_array_equal(res.raw_triggers, ref_gpu.raw_triggers)
    np.testing.assert_array_equal(res.fixed_triggers, ref_gpu.fixed_triggers)
    np.testing.assert_array_equal(res.sc_freq.view(np.uint32), ref_gpu.sc_freq.view(np.uint32))
    assert len(res.frames) == len(ref_gpu.frames) > 20
    for k in FRAME_KEYS:
        a, b = res.frames[k], ref_gpu.frames[k]
        np.testing.assert_array_equal(a.view(np.uint64) if a.dtype.kind == "f" else a,
                                      b.view(np.uint64) if b.dtype.kind == "f" else b, err_msg=k)
    for k in TB_KEYS:
        np.testing.assert_array_equal(res.tbs[k], ref_gpu.tbs[k], err_msg=k)
    assert torch.equal(res.payload, ref_gpu.payload)
    # and the oracle
    ref = oracle.rx_stream(ocfg, code, x, feedback=oracle_feedback(oracle, pcfg, rx), want=("frames", "trig"))
    np.testing.assert_array_equal(res.raw_triggers, np.nonzero(ref["trig_raw"])[0])
    np.testing.assert_array_equal(res.fixed_triggers, np.nonzero(ref["trig_fixed"])[0])
    assert [(t["tb_no"], t["crc_ok"], t["avg_it"]) for t in res.tbs] == [(t["tb_no"], t["crc_ok"], t["avg_it"]) for t in ref["tbs"]]
    np.testing.assert_array_equal(res.payload.cpu().numpy(), ref["user_bytes"])
    if drop or jitter:
        assert not np.array_equal(res.raw_triggers, res.fixed_triggers)   # frame_detect_bb did change the list


@pytest.mark.gpu
def test_pipelined_feed_with_carried_triggers(oracle, dtl, cuda, split_env):
    """Chunked feed(): every call is large enough to be pipelined and starts with triggers carried over."""
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=64, sync_threshold=0.95)
    user, x = _jittered(oracle, ocfg, code, 2, 30000, 14.0, seed=77, drop_every=9, jitter=2)
    xt = torch.from_numpy(x).to(cuda)
    batch = dtl.ofdm_receiver(pcfg).work(xt)
    rx = dtl.ofdm_receiver(pcfg)
    n = xt.numel()
    cuts = [n // 3 + 411, 2 * n // 3 + 1007]
    assert min(np.diff([0] + cuts + [n])) > 20000
    _assert_same(batch, _feed_all(rx, xt, cuts))


def _same_result(a, b):
    import torch
    np.testing.assert_array_equal(a.raw_triggers, b.raw_triggers)
    np.testing.assert_array_equal(a.fixed_triggers, b.fixed_triggers)
    np.testing.assert_array_equal(a.sc_freq.view(np.uint32), b.sc_freq.view(np.uint32))
    assert len(a.frames) == len(b.frames)
    for k in FRAME_KEYS:
        x, y = a.frames[k], b.frames[k]
        np.testing.assert_array_equal(x.view(np.uint64) if x.dtype.kind == "f" else x, y.view(np.uint64) if y.dtype.kind == "f" else y, err_msg=k)
    for k in TB_KEYS:
        np.testing.assert_array_equal(a.tbs[k], b.tbs[k], err_msg=k)
    assert torch.equal(a.payload, b.payload)
    if a.llr is not None:
        assert torch.equal(a.llr.view(torch.int32), b.llr.view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("split", [False, True])
def test_submit_wait_equals_work_batch_by_batch(oracle, dtl, cuda, monkeypatch, split):
    """dtl_rx_submit / dtl_rx_wait (two batches in flight on the handle's two lanes -- two receivers, two host threads, two
    streams): five different batches through the two-deep queue return, one by one, exactly what work() returns for each -- triggers,
    frame records, block records, soft bits and bytes -- with the single-range and the four-range sync stage."""
    import torch
    if split:
        monkeypatch.setenv("DTL_RX_SPLIT_MIN", "20000")
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=64, sync_threshold=0.95)
    batches = []
    for i, (cnst, esn0, nb) in enumerate([(2, 13.0, 12000), (4, 24.0, 20000), (1, 20.0, 4000), (3, 21.0, 16000), (2, 14.0, 9000)]):
        user, x = _jittered(oracle, ocfg, code, cnst, nb, esn0, seed=500 + i, drop_every=(6 if i % 2 else 0), jitter=i % 3)
        batches.append(torch.from_numpy(x).to(cuda))
    want = [dtl.ofdm_receiver(pcfg).work(x, want=("llr",)) for x in batches]
    rx = dtl.ofdm_receiver(pcfg)
    got = []
    rx.submit(batches[0], want=("llr",))
    for i in range(len(batches)):
        if i + 1 < len(batches):
            rx.submit(batches[i + 1], want=("llr",))
        got.append(rx.wait())
    for a, b in zip(got, want):
        _same_result(a, b)
    assert sum(int((r.tbs["crc_ok"] == 1).sum()) for r in got) > 100
    # the same handle goes on with plain calls
    _same_result(rx.work(batches[1], want=("llr",)), want[1])


@pytest.mark.gpu
def test_submit_wait_queue_rules(oracle, dtl, cuda):
    import torch
    from gr_dtl_amd import _lib
    pcfg, ocfg, code = make_pair(oracle, dtl)
    user, hdrs, x = make_stream(oracle, ocfg, code, 2, 3000, 13.0, seed=9)
    xt = torch.from_numpy(x).to(cuda)
    rx = dtl.ofdm_receiver(pcfg)
    with pytest.raises(_lib.DtlError) as e:
        rx.wait()
    assert e.value.code == _lib.ERR_INVALID_ARG
    rx.submit(xt)
    rx.submit(xt)
    with pytest.raises(_lib.DtlError) as e:
        rx.submit(xt)                                  # two are waiting already
    assert e.value.code == _lib.ERR_CAPACITY
    with pytest.raises(_lib.DtlError):
        rx.work(xt)                                    # a plain call while batches wait is refused
    a, b = rx.wait(), rx.wait()
    assert torch.equal(a.payload, b.payload) and a.payload.numel() > 0
    ref = rx.work(xt)
    assert torch.equal(ref.payload, a.payload)
    # an empty batch and a handle destroyed with a batch waiting
    rx.submit(xt[:0])
    assert rx.wait().n_payload_bytes == 0
    rx.submit(xt)
    del rx
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_handle_bound_to_a_device(oracle, dtl, cuda):
    """dtl_rx_cfg.device: a handle created for device 0 behaves like one that takes the caller's current device, leaves
    the caller's current device alone, and an index the machine does not have is a status at the first call."""
    import dataclasses
    import torch
    from gr_dtl_amd import _lib
    pcfg, ocfg, code = make_pair(oracle, dtl)
    user, hdrs, x = make_stream(oracle, ocfg, code, 2, 3000, 13.0, seed=21)
    xt = torch.from_numpy(x).to(cuda)
    want = dtl.ofdm_receiver(pcfg).work(xt)
    bound = dtl.ofdm_receiver(dataclasses.replace(pcfg, device=0))
    got = bound.work(xt)
    assert torch.equal(got.payload, want.payload) and got.payload.numel() > 0
    assert torch.cuda.current_device() == 0
    far = dtl.ofdm_receiver(dataclasses.replace(pcfg, device=torch.cuda.device_count() + 3))
    with pytest.raises(_lib.DtlError) as e:
        far.work(xt)
    assert e.value.code in (_lib.ERR_HIP, _lib.ERR_INVALID_ARG, _lib.ERR_NO_DEVICE), e.value
    assert bound.work(xt).n_payload_bytes == want.n_payload_bytes          # the failed handle disturbed nobody


@pytest.mark.gpu
def test_a_lanes_failure_arrives_at_wait_with_its_text(oracle, dtl, cuda):
    """A batch that fails on its lane's thread: dtl_rx_wait returns the status and dtl_last_error() the text (which the
    library keeps per thread), for both lanes; the handle's queue is empty afterwards."""
    import torch
    from gr_dtl_amd import _lib
    pcfg = dtl.ofdm_adaptive_rx_config(frame_length=4)
    pcfg.occupied_carriers = (tuple(pcfg.occupied_carriers[0])[4:-4],)        # 96 header bits do not fit: refused at the call
    pcfg.fec_codes = (("fec_1", "n_0300_k_0152_gap_03.alist"),)
    pcfg.mcs = ((-100000, (dtl.constellation_type_t.BPSK, "fec_1")),)
    rx = dtl.ofdm_receiver(pcfg)
    x = torch.zeros(5000, dtype=torch.complex64, device=cuda)
    rx.submit(x)
    rx.submit(x)
    for _ in range(2):
        with pytest.raises(_lib.DtlError) as e:
            rx.wait()
        assert e.value.code == _lib.ERR_INVALID_ARG and "header" in str(e.value)
    with pytest.raises(_lib.DtlError) as e:
        rx.wait()
    assert "no batch" in str(e.value)


@pytest.mark.gpu
def test_many_batches_through_both_lanes_in_order(oracle, dtl, cuda):
    """Twelve batches of three kinds through the two lanes, two always in flight: every result is its own batch's (the
    order of wait() is the order of submit()), whichever lane and however long a batch takes."""
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=64, sync_threshold=0.95)
    kinds = []
    for i, (cnst, esn0, nb) in enumerate([(2, 13.0, 30000), (4, 24.0, 3000), (1, 20.0, 9000)]):    # long, short, middle
        user, hdrs, x = make_stream(oracle, ocfg, code, cnst, nb, esn0, seed=900 + i)
        xt = torch.from_numpy(x).to(cuda)
        kinds.append((xt, dtl.ofdm_receiver(pcfg).work(xt)))
    order = [0, 1, 1, 2, 0, 0, 2, 1, 2, 2, 0, 1]
    rx = dtl.ofdm_receiver(pcfg)
    got = []
    rx.submit(kinds[order[0]][0])
    for j in range(len(order)):
        if j + 1 < len(order):
            rx.submit(kinds[order[j + 1]][0])
        got.append(rx.wait())
    for j, r in enumerate(got):
        _same_result(r, kinds[order[j]][1])


@pytest.mark.gpu
def test_lanes_under_a_random_schedule(oracle, dtl, cuda):
    """400 operations drawn at random -- submit (when fewer than two wait), wait, a plain call when nothing waits, now and
    then a new handle while the old one still has batches waiting -- against the byte counts and block records each batch
    has on its own: no deadlock, no result of another batch, no state carried from a batch into the next."""
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=64, sync_threshold=0.95)
    kinds = []
    for i, (cnst, esn0, nb) in enumerate([(2, 13.0, 6000), (4, 24.0, 2000), (1, 20.0, 500), (3, 21.0, 3000)]):
        user, hdrs, x = make_stream(oracle, ocfg, code, cnst, nb, esn0, seed=1300 + i)
        xt = torch.from_numpy(x).to(cuda)
        r = dtl.ofdm_receiver(pcfg).work(xt)
        kinds.append((xt, r.n_payload_bytes, [(int(t["tb_no"]), int(t["crc_ok"])) for t in r.tbs], r.payload.cpu().numpy()))
    kinds.append((kinds[0][0][:0], 0, [], np.zeros(0, np.uint8)))                      # an empty batch
    rng = np.random.default_rng(44)
    rx = dtl.ofdm_receiver(pcfg)
    waiting = []
    checked = 0
    log = []
    for step in range(400):
        op = rng.random()
        if op < 0.02:
            rx = dtl.ofdm_receiver(pcfg)              # the old handle goes away with whatever it still had
            waiting = []
            log.append("new")
            continue
        elif (op < 0.55 and len(waiting) < 2) or not waiting:
            if not waiting and op > 0.9:
                k = int(rng.integers(len(kinds)))
                r = rx.work(kinds[k][0])
                log.append("work%d" % k)
            else:
                k = int(rng.integers(len(kinds)))
                rx.submit(kinds[k][0])
                waiting.append(k)
                log.append("sub%d" % k)
                continue
        else:
            k = waiting.pop(0)
            r = rx.wait()
            log.append("wait%d" % k)
        assert r.n_payload_bytes == kinds[k][1], (step, k, " ".join(log[-14:]))
        assert [(int(t["tb_no"]), int(t["crc_ok"])) for t in r.tbs] == kinds[k][2], (step, k)
        if step % 7 == 0:
            np.testing.assert_array_equal(r.payload.cpu().numpy(), kinds[k][3])
        checked += 1
    assert checked > 150
    while waiting:
        k = waiting.pop(0)
        assert rx.wait().n_payload_bytes == kinds[k][1]
