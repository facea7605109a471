"""Two receivers driven from two host threads on two streams at once (a flowgraph with two receive chains, or a
caller that overlaps one batch's host work with another's kernels): every handle's results equal those of the same
calls made one after the other.  The library's process-wide state (kernel attributes, CRC / FFT tables built on first
use) is guarded; everything else belongs to a handle."""
import threading

import numpy as np
import pytest

from helpers import make_pair, make_stream


def _snapshot(res):
    return (np.array(res.raw_triggers), np.array(res.fixed_triggers), res.frames.copy(), res.tbs.copy(),
            res.payload.cpu().numpy().copy(), res.llr.cpu().numpy().copy())


@pytest.mark.gpu
def test_two_receivers_in_two_threads(oracle, dtl, cuda):
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl)
    streams_np = [make_stream(oracle, ocfg, code, cnst, 9000, esn0, seed=900 + cnst)[2] for cnst, esn0 in ((4, 23.0), (2, 13.0))]
    xs = [torch.from_numpy(x).to(cuda) for x in streams_np]
    # one after the other, one handle each
    want = []
    for x in xs:
        rx = dtl.ofdm_receiver(pcfg)
        want.append(_snapshot(rx.work(x, want=("llr",))))
    # at once: fresh handles (first use of every lazily built table happens under contention), own streams
    got = [[None] * 6 for _ in xs]
    errors = []
    go = threading.Barrier(len(xs))

    def worker(j):
        try:
            torch.cuda.set_device(cuda)
            rx = dtl.ofdm_receiver(pcfg)
            with torch.cuda.stream(torch.cuda.Stream(device=cuda)):
                go.wait()
                for k in range(6):
                    got[j][k] = _snapshot(rx.work(xs[j], want=("llr",)))
        except Exception as e:      # noqa: BLE001 -- reported below, in the main thread
            errors.append((j, repr(e)))

    ts = [threading.Thread(target=worker, args=(j,)) for j in range(len(xs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for j in range(len(xs)):
        assert len(want[j][3]) > 0 and (want[j][3]["crc_ok"] == 1).any()
        for k in range(6):
            for a, b in zip(got[j][k], want[j]):
                if a.dtype.kind == "f":
                    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
                else:
                    assert np.array_equal(a, b)
