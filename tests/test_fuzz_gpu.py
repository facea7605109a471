"""Randomised differential tests: the HIP path against the oracle on seeded random inputs.  Fixed seeds: a failure reproduces.

  test_random_scenario_matches_oracle      whole receive chain; FFT length, frame length, constellation sequence, Es/N0 or
                                           noise-free, carrier frequency offset, sync threshold, frame_detect_bb on / off,
                                           FEC code or none, idle gaps, multipath, erased / cut frames, clock slips, stream
                                           cut short; every stage compared as in test_rx_gpu, then the same stream fed in
                                           randomly cut calls (bit for bit against the batch call)
  test_random_scenario_with_carrier_plans  the same with 512 / 1024-point FFTs, random carrier plans (ragged pilot sets,
                                           frames that hold no whole number of bytes), a wrapping frame counter, the
                                           decoder's iteration limit and sigma
  test_random_tx_matches_oracle            the batched transmitter, bit for bit
  test_random_ldpc_inputs_match_oracle     the decoder alone on any float (inf, NaN, zeros, denormals, shortened words)
  test_random_demap_inputs_match_oracle    the demapper alone on any float and any noise power
  test_random_equalizer_inputs_match_oracle  channel estimator and equaliser alone on vectors from no transmitter

Environment: DTL_FUZZ_FIRST / DTL_FUZZ_LAST, DTL_FUZZ_PLAN_FIRST / DTL_FUZZ_PLAN_LAST (seed ranges of the two scenario
families; defaults 0..399 and 0..239), DTL_FUZZ_AUX (multiplier of the block-level seed counts), DTL_FUZZ_SPECIALS /
DTL_FUZZ_NO_SPECIALS (special values mixed into the equaliser inputs)."""
import os

import numpy as np
import pytest

from helpers import ES, make_pair
from test_rx_gpu import _compare, _run_both
from test_stream_gpu import _assert_same, _feed_all

AUX = int(os.environ.get("DTL_FUZZ_AUX", "1"))     # multiplies the seed counts of the block-level tests below
CODES = ("n_0300_k_0152_gap_03.alist", "n_0100_k_0023_gap_10.alist", "n_0100_k_0027_gap_04.alist")


def _scenario(oracle, dtl, rng, plans=False, rotating_sets=False):
    fft_len = int(rng.choice([64, 64, 64, 128, 128, 256, 256, 512, 1024] if plans else [64, 64, 128, 256]))
    frame_length = int(rng.integers(1, 21)) if fft_len == 64 else int(rng.integers(2, 13)) if fft_len <= 256 else int(rng.integers(2, 6))
    fec = rng.choice([None, *CODES]) if fft_len == 64 else rng.choice(CODES)
    if rotating_sets:
        fft_len = 64
    thr = float(rng.choice([0.95, 0.95, 0.9, 0.97]))
    knobs = {}
    if plans:   # the decoder's iteration limit and its channel sigma (never set by the reference's flowgraphs); not header
        # scrambling: the reference's parser does not undo the formatter's mask (packet_header.cc:194-197 vs :231-312) --
        # with it no header passes its CRC, in the reference, the oracle and the product alike
        knobs = dict(max_it=int(rng.choice([15, 15, 5, 30])), bp_sigma=float(rng.choice([0.0, 0.0, 1.0, 2.5])))
    carriers = None
    if plans and fft_len == 64 and rng.integers(0, 3) == 0:
        # a carrier plan of its own: the 52 carriers the sync words cover, 2 .. 4 of them pilots anywhere (48 .. 50 data
        # carriers: the header needs 48 per symbol), one or two pilot sets (the second with fewer pilots: ragged
        # tables), fresh pilot symbols
        used = [c for c in range(-26, 27) if c]
        pil = sorted(int(c) for c in rng.choice(used, int(rng.choice([2, 3, 4])), replace=False))
        occ = tuple(c for c in used if c not in pil)
        sets = (tuple(pil),) if rng.integers(0, 2) else (tuple(pil), tuple(pil[:len(pil) // 2]))
        rows = tuple(tuple(complex(v) for v in rng.choice([-1.0, 1.0], len(sets[i % len(sets)])))
                     for i in range(len(sets) * int(rng.choice([1, 2, 3]))))     # row i belongs to pilot set i mod n_sets
        carriers = ((occ,), sets, rows)
    if rotating_sets:
        # Several data-carrier sets of DIFFERENT sizes, taken in turn symbol by symbol (ofdm_serializer_vcc /
        # ofdm_carrier_allocator_cvc: symbol i -> set i mod n_sets).  The header symbols' sets must hold the formatter's
        # header_len x len(occupied_carriers[0]) bits (see RxConfig::check_header_plan): with FEC (two header symbols) the
        # first two sets have one size and a third differs; without FEC the second set already may.  frame_length is a
        # multiple of the number of sets, so that the frame capacity does not depend on where the rotation starts (the
        # reference's transmitter starts the payload at set header_len, its receiver's serializer at set 1).
        used = [c for c in range(-26, 27) if c]
        pil = (-21, -7, 7, 21)
        base = [c for c in used if c not in pil]                    # 48 data carriers
        def drop(k):
            gone = set(int(c) for c in rng.choice(base, k, replace=False))
            return tuple(c for c in base if c not in gone)
        if fec is None:
            occ = (tuple(base), drop(int(rng.integers(1, 9))))
        else:
            occ = (tuple(base), tuple(base[::-1]) if rng.integers(0, 2) else tuple(base), drop(int(rng.integers(1, 9))))
        frame_length = len(occ) * int(rng.integers(1, 7))
        cfg0 = dtl.ofdm_adaptive_rx_config()
        carriers = (occ, cfg0.pilot_carriers, cfg0.pilot_symbols)
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=fft_len, frame_length=frame_length, fec=None if fec is None else str(fec),
                                 sync_threshold=thr, use_sync_correct=bool(rng.integers(0, 2)),
                                 llr_dist_mode=int(rng.integers(0, 2)), arith=str(rng.choice(["gr_exact", "gr_exact", "fast"])),
                                 carriers=carriers, **knobs)
    n_seg = int(rng.integers(1, 4))
    impair = bool(rng.integers(0, 2))
    channel = None
    if rng.integers(0, 3) == 0:                                      # a short multipath channel inside the cyclic prefix
        taps = int(rng.integers(2, max(3, pcfg.cp_len // 2)))
        channel = (rng.normal(0, 1, taps) + 1j * rng.normal(0, 1, taps)) * np.exp(-0.7 * np.arange(taps))
        channel[0] = 1.0
        channel = (channel / np.sqrt(np.sum(np.abs(channel) ** 2))).astype(np.complex64)
    parts, frame_no, tb_no = [], (int(rng.integers(0, 4096)) if plans else 0), 0   # (plans: the 12-bit frame counter wraps)
    worst = 0.0
    gap = int(rng.choice([0, 0, 0, 7, 64]))
    for _ in range(n_seg):
        cnst = int(rng.integers(1, 5))
        per_frame = max(ocfg.frame_capacity * cnst // 8, 8)
        user = rng.integers(0, 256, int(rng.integers(1, 6 * per_frame)), dtype=np.uint8)
        if code is not None and (max(1, 1 + (ocfg.frame_capacity * cnst // 8 * 8) // code.n) * code.k) // 8 <= 4:
            cnst = 4                                        # this code's transport block holds no payload at this frame size
        if code is not None:
            syms, hdrs = oracle.fec_frames_build(code, 1, cnst, ocfg.frame_capacity, user, first_frame_no=frame_no)
        else:
            syms, hdrs = oracle.nofec_frames_build(cnst, ocfg.frame_capacity, user, first_frame_no=frame_no)
        frame_no = (frame_no + len(hdrs)) & 0xFFF
        iq = oracle.tx_frames(ocfg, syms, hdrs)
        if gap:
            iq = np.concatenate([iq, np.zeros((iq.shape[0], gap), np.complex64)], axis=1)
        rows = [r for r in iq]
        if impair:
            for j in range(len(rows)):
                u = rng.integers(0, 24)
                if u == 0:
                    rows[j] = np.zeros_like(rows[j])                 # a frame that never arrives (idle air)
                elif u == 1:
                    rows[j] = rows[j][:-1]                           # the transmitter's clock slips one sample
                elif u == 2:
                    rows[j] = np.concatenate([rows[j], rows[j][-1:]])
                elif u == 3 and len(rows) > 2:
                    rows[j] = rows[j][:0]                            # a frame cut out of the stream entirely
        parts.append(np.concatenate(rows) if rows else np.zeros(0, np.complex64))
        worst = max(worst, ES[cnst] / {1: 1.0, 2: 1.0, 3: 3.0, 4: 6.0}[cnst])
    x = np.concatenate([np.zeros(int(rng.integers(0, 300)), np.complex64)] + parts +
                       [np.zeros(int(rng.integers(0, ocfg.frame_samples + 1)), np.complex64)])
    if channel is not None:
        x = np.convolve(x, channel).astype(np.complex64)[:x.size]
    esn0 = float(rng.uniform(17.0, 30.0))
    nv = fft_len * 1.0 / 10 ** (esn0 / 10)
    if rng.integers(0, 4) == 0:
        nv = 0.0                                                                   # a noise-free scenario now and then
    eps = float(rng.choice([0.0, 0.0, rng.uniform(-0.45, 0.45), rng.uniform(-3.0, 3.0)]))
    if eps:
        n = np.arange(x.size, dtype=np.float64)
        x = (x * np.exp(2j * np.pi * eps * n / fft_len)).astype(np.complex64)
    if nv:
        x = (x + (rng.normal(0, np.sqrt(nv / 2), x.size) + 1j * rng.normal(0, np.sqrt(nv / 2), x.size))).astype(np.complex64)
    if rng.integers(0, 5) == 0:
        x = x[:int(x.size * rng.uniform(0.5, 0.99))]                               # the capture ends inside a frame
    return pcfg, ocfg, code, np.ascontiguousarray(x), dict(fft_len=fft_len, frame_length=frame_length, fec=fec, thr=thr,
                                                           gap=gap, esn0=esn0, eps=eps, n=x.size, impair=impair,
                                                           channel=None if channel is None else channel.size)


def _assert_production_form_equals_general_form(batch, res, has_fec):
    """`res` (compared with the oracle stage by stage) asked for the equaliser's rows and therefore ran the general form of
    phase B (frame_eq_kernel<false> + demap_kernel); `batch` is what a caller gets who asks for none: the staged frame
    equaliser (no global memory inside an equaliser step) where the configuration allows it.  Same arithmetic:
    soft bits, serializer output, block records and bytes must be bit-identical."""
    assert np.array_equal(batch.tbs, res.tbs)
    np.testing.assert_array_equal(batch.payload.cpu().numpy(), res.payload.cpu().numpy())
    for name in (("llr", "soft_syms") if has_fec else ("soft_syms",)):   # (no soft bits are formed without a code)
        a, b = getattr(batch, name), getattr(res, name)
        if a is None or b is None:
            assert (a is None or a.numel() == 0) and (b is None or b.numel() == 0), name
            continue
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert a.shape == b.shape, name
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32), err_msg=name)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("DTL_FUZZ_FIRST", "0")), int(os.environ.get("DTL_FUZZ_LAST", "400")))))
def test_random_scenario_matches_oracle(oracle, dtl, cuda, seed):
    rng = np.random.default_rng(9000 + seed)
    pcfg, ocfg, code, x, info = _scenario(oracle, dtl, rng)
    rx, res, ref = _run_both(oracle, dtl, cuda, pcfg, ocfg, code, x, max_frames=x.size // 64 + 16)
    _compare(rx, res, ref, ocfg, code is not None)
    # the same stream cut at random points and fed call by call: bit-identical to the batch call
    import torch
    xt = torch.from_numpy(x).to(cuda)
    n_cuts = int(rng.choice([1, 3, 12, 40]))
    cuts = sorted(set(int(c) for c in rng.integers(1, max(x.size, 2), n_cuts))) if x.size > 1 else []
    batch, streamed = dtl.ofdm_receiver(pcfg).work(xt, want=("llr", "soft_syms")), _feed_all(dtl.ofdm_receiver(pcfg), xt, cuts)
    _assert_production_form_equals_general_form(batch, res, code is not None)
    if seed % 3 == 0:        # ... and the opt-in one-kernel form of phase B (soft demapper in the equaliser's epilogue)
        os.environ["DTL_RX_FUSED_PHASE_B"] = "1"
        try:
            fused = dtl.ofdm_receiver(pcfg).work(xt, want=("llr", "soft_syms"))
        finally:
            os.environ.pop("DTL_RX_FUSED_PHASE_B", None)
        _assert_production_form_equals_general_form(fused, res, code is not None)
    if pcfg.use_sync_correct and not np.array_equal(streamed[4], batch.fixed_triggers):
        # frame_detect_bb looks ahead to the end of the buffer it is given (insertion of a missing trigger, the
        # "fits into this buffer" test of a correction): as in the reference its output depends on the cuts
        np.testing.assert_array_equal(streamed[3], batch.raw_triggers)
        return
    _assert_same(batch, streamed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("DTL_FUZZ_PLAN_FIRST", "0")), int(os.environ.get("DTL_FUZZ_PLAN_LAST", "240")))))
def test_random_scenario_with_carrier_plans(oracle, dtl, cuda, seed):
    """As above, with 512- and 1024-point FFTs, a frame counter that starts anywhere (and wraps), and a third of the
    64-carrier scenarios on a random carrier plan (48 .. 50 data carriers, pilots anywhere, ragged pilot sets: frame
    capacities that hold no whole number of bytes)."""
    rng = np.random.default_rng(123000 + seed)
    pcfg, ocfg, code, x, info = _scenario(oracle, dtl, rng, plans=True)
    rx, res, ref = _run_both(oracle, dtl, cuda, pcfg, ocfg, code, x, max_frames=x.size // 64 + 16)
    _compare(rx, res, ref, ocfg, code is not None)
    import torch
    batch = dtl.ofdm_receiver(pcfg).work(torch.from_numpy(x).to(cuda), want=("llr", "soft_syms"))
    _assert_production_form_equals_general_form(batch, res, code is not None)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("DTL_FUZZ_SETS_FIRST", "0")), int(os.environ.get("DTL_FUZZ_SETS_LAST", "60")))))
def test_random_scenario_with_rotating_carrier_sets(oracle, dtl, cuda, seed):
    """Data-carrier sets of different sizes taken in turn (two without FEC, three with): header bits (row stride =
    dtl_rx_result.hdr_bits), serializer output, soft bits and every record against the oracle.  With FEC the reference's
    transmitter and receiver rotate the payload's sets from different starts (allocator: header_len, serializer: 1,
    python/dtl/ofdm_receiver.py:175-180), so the payload decodes to nothing -- on both sides, identically."""
    rng = np.random.default_rng(456000 + seed)
    pcfg, ocfg, code, x, info = _scenario(oracle, dtl, rng, rotating_sets=True)
    assert len(pcfg.occupied_carriers) >= 2 and len({len(s) for s in pcfg.occupied_carriers}) >= 2
    rx, res, ref = _run_both(oracle, dtl, cuda, pcfg, ocfg, code, x, max_frames=x.size // 64 + 16)
    assert res.hdr_bits is None or res.hdr_bits.shape[1] == rx.header_len * len(pcfg.occupied_carriers[0])
    _compare(rx, res, ref, ocfg, code is not None)
    import torch
    xt = torch.from_numpy(x).to(cuda)
    batch = dtl.ofdm_receiver(pcfg).work(xt, want=("llr", "soft_syms"))
    _assert_production_form_equals_general_form(batch, res, code is not None)
    cuts = sorted(set(int(c) for c in rng.integers(1, max(x.size, 2), 5))) if x.size > 1 else []
    streamed = _feed_all(dtl.ofdm_receiver(pcfg), xt, cuts)
    if pcfg.use_sync_correct and not np.array_equal(streamed[4], batch.fixed_triggers):
        np.testing.assert_array_equal(streamed[3], batch.raw_triggers)
        return
    _assert_same(batch, streamed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(150 * AUX)))
def test_random_tx_matches_oracle(oracle, dtl, cuda, seed):
    """The batched transmitter against the oracle's: framing, header fields and on-air samples bit for bit."""
    import torch
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    rng = np.random.default_rng(77000 + seed)
    fft_len = int(rng.choice([64, 64, 128, 256, 512]))
    frame_length = int(rng.integers(1, 21)) if fft_len == 64 else int(rng.integers(2, 9))
    fec = rng.choice([None, *CODES])
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=fft_len, frame_length=frame_length, fec=None if fec is None else str(fec))
    cnst = int(rng.integers(1, 5))
    if code is not None and (max(1, 1 + (ocfg.frame_capacity * cnst // 8 * 8) // code.n) * code.k) // 8 <= 4:
        cnst = 4
    per_frame = max(ocfg.frame_capacity * cnst // 8, 8)
    user = rng.integers(0, 256, int(rng.choice([1, 2, per_frame - 1, per_frame, per_frame + 1, int(rng.integers(1, 12 * per_frame))])),
                        dtype=np.uint8)
    first = int(rng.integers(0, 4096))
    if code is not None:
        ref_syms, ref_hdrs = oracle.fec_frames_build(code, 1, cnst, ocfg.frame_capacity, user, first_frame_no=first)
    else:
        ref_syms, ref_hdrs = oracle.nofec_frames_build(cnst, ocfg.frame_capacity, user, first_frame_no=first)
    ref_iq = oracle.tx_frames(ocfg, ref_syms, ref_hdrs)
    tx = ofdm_transmitter(pcfg)
    args = (cnst, "fec_1") if code is not None else (cnst,)
    iq, hdrs, syms = tx.work_device(torch.from_numpy(user).to(cuda), *args, first_frame_no=first, want_syms=True)
    assert len(hdrs) == len(ref_hdrs)
    keys = ("payload_len", "frame_no", "cnst", "fb_cnst") + (("tb_no", "fec_fb", "tb_offset", "fec", "tb_payload") if code is not None else ())
    for k in keys:
        np.testing.assert_array_equal(hdrs[k], [h[k] for h in ref_hdrs], err_msg=k)
    np.testing.assert_array_equal(syms.cpu().numpy(), ref_syms)
    got = np.ascontiguousarray(iq.cpu().numpy())
    assert got.shape == ref_iq.shape
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(ref_iq).view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(120 * AUX)))
def test_random_ldpc_inputs_match_oracle(oracle, dtl, cuda, seed):
    """The belief-propagation decoder alone on soft bits no demapper would produce: random scale, sign errors, runs of
    shortened positions, +-inf, NaN, zeros, denormals, exponents near the "tame" bound of the shortened division
    sequence (|llr| ~ 500 ln 2), any iteration limit; bits and iteration counts exact in both arithmetics."""
    import torch
    from conftest import CODES as CODE_DIR
    rng = np.random.default_rng(31000 + seed)
    alist = os.path.join(CODE_DIR, str(rng.choice(CODES)))
    code = oracle.Ldpc(alist)
    ncw = int(rng.choice([1, 2, 63, 64, 65, 257, int(rng.integers(1, 6000))]))
    data = rng.integers(0, 2, (ncw, code.k)).astype(np.uint8)
    cws = np.stack([code.encode(d) for d in data]).astype(np.float64)
    scale = float(10 ** rng.uniform(-1.5, 2.9))
    noise = float(rng.uniform(0.2, 1.4))
    llr = ((2 * cws - 1) + rng.normal(0, noise, cws.shape)) * scale
    u = rng.random(cws.shape)
    p = rng.uniform(0, 0.02, 6) * (rng.random(6) < 0.5)
    llr = np.where(u < p[0], np.inf, llr)
    llr = np.where((u >= 0.1) & (u < 0.1 + p[1]), -np.inf, llr)
    llr = np.where((u >= 0.2) & (u < 0.2 + p[2]), np.nan, llr)
    llr = np.where((u >= 0.3) & (u < 0.3 + p[3]), 0.0, llr)
    llr = np.where((u >= 0.4) & (u < 0.4 + p[4]), 1e-41 * np.sign(llr), llr)
    llr = np.where((u >= 0.5) & (u < 0.5 + p[5]), rng.uniform(330, 360, cws.shape) * np.sign(llr), llr)
    if rng.integers(0, 3) == 0:                                       # shortened codewords: a run of -15
        cut = int(rng.integers(code.n - code.k, code.n))
        llr[:, cut:] = -15.0
    llr = llr.astype(np.float32)
    max_it = int(rng.choice([1, 1, 2, 8, 15, 15, 15, 50]))
    for arith, oa in (("gr_exact", oracle.GR_EXACT), ("fast", oracle.FAST)):
        with np.errstate(all="ignore"):
            ref_bits, ref_nit = code.decode_batch(llr, max_it=max_it, n_threads=8, arith=oa)
        dec = dtl.ldpc_dec(alist, max_it=max_it, arith=arith)
        bits, nit = dec.decode_batch(torch.from_numpy(llr).to(cuda))
        np.testing.assert_array_equal(nit.cpu().numpy(), ref_nit, err_msg=arith)
        np.testing.assert_array_equal(bits.cpu().numpy(), ref_bits, err_msg=arith)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(80 * AUX)))
def test_random_demap_inputs_match_oracle(oracle, dtl, cuda, seed):
    """The soft demapper alone on symbols no equaliser should produce (huge, tiny, denormal, infinite, NaN, exactly on a
    constellation point or a decision boundary) and any noise power, every constellation, both distance modes, both
    arithmetics: the same bits as the oracle (NaN in the same places)."""
    import torch
    from gr_dtl_amd import stages
    rng = np.random.default_rng(52000 + seed)
    cnst = int(rng.integers(1, 5))
    n = int(rng.choice([1, 63, 64, 65, 1000, 4099]))
    scale = float(10 ** rng.uniform(-3, 3))
    z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * scale
    u = rng.random(n)
    special = np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 1e-41, -1e-41, 3e38, -3e38, 1.0, -1.0, 0.5, 2.0 / np.sqrt(10)])
    re = np.where(u < 0.1, rng.choice(special, n), z.real)
    im = np.where((u > 0.05) & (u < 0.15), rng.choice(special, n), z.imag)
    syms = (re + 1j * im).astype(np.complex64)
    noise = rng.choice([None, 0.0, 1e-42, 1e-30, 1e-6, 0.01, 0.3, 1.0, 50.0, 1e20, np.inf])
    dist_mode = int(rng.integers(0, 2))
    for arith, oa in (("gr_exact", oracle.GR_EXACT), ("fast", oracle.FAST)):
        got = stages.soft_demap(torch.from_numpy(syms).to(cuda), cnst, noise, dist_mode, arith=arith).cpu().numpy()
        want = oracle.soft_demap(cnst, syms, noise, dist_mode, arith=oa)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (cnst, dist_mode, noise, arith, int((~same).sum()), syms[np.nonzero(~same)[0][:4] // max(1, got.size // n)],
                            got[~same][:4], want[~same][:4])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(60 * AUX)))
def test_random_equalizer_inputs_match_oracle(oracle, dtl, cuda, seed):
    """Channel estimator and payload frame equaliser alone on vectors that come from no transmitter: Gaussian carriers of
    any scale with exact zeros and NaN mixed in, taps that are zero on some carriers (division by zero inside the
    decision-directed recursion), every constellation and carrier offset.  Zeros compare without their sign."""
    import torch
    rng = np.random.default_rng(64000 + seed)
    fft_len = int(rng.choice([64, 64, 128, 256]))
    L = int(rng.integers(1, 13))
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=fft_len, frame_length=L)
    rx = dtl.ofdm_receiver(pcfg)
    n = int(rng.integers(1, 9))
    # Denormals, |x| > 2^62 and infinities are refused by these entry points (the next test).
    special = np.array([0.0, -0.0, np.nan], np.float64)
    if os.environ.get("DTL_FUZZ_SPECIALS"):
        special = np.array([float(v) for v in os.environ["DTL_FUZZ_SPECIALS"].split(",")])

    def garbage(shape, p_special):
        z = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 10 ** rng.uniform(-2, 2)
        u = rng.random(shape)
        re = np.where(u < p_special, rng.choice(special, shape), z.real)
        im = np.where((u > p_special / 2) & (u < 1.5 * p_special), rng.choice(special, shape), z.imag)
        return (re + 1j * im).astype(np.complex64)

    p = float(rng.choice([0.0, 0.002, 0.05]))
    if os.environ.get("DTL_FUZZ_NO_SPECIALS"):
        p = 0.0
    sync = garbage((n, 2, fft_len), p)
    off, taps = rx.chanest(torch.from_numpy(sync).to(cuda))
    off_h, taps_h = off.cpu().numpy(), taps.cpu().numpy()
    with np.errstate(all="ignore"):
        for j in range(n):
            o_ref, t_ref = oracle.chanest(ocfg, sync[j])
            assert off_h[j] == o_ref
            same = (taps_h[j].view(np.float32) == t_ref.view(np.float32)) | (np.isnan(taps_h[j].view(np.float32)) & np.isnan(t_ref.view(np.float32)))
            assert same.all(), ("taps", j, int((~same).sum()))
    frames = garbage((n, L, fft_len), p)
    taps_in = garbage((n, fft_len), float(rng.choice([0.0, 0.02])) if p else 0.0)
    cn = [int(c) for c in rng.integers(1, 5, n)]
    of = [int(o) for o in rng.choice([-4, -2, 0, 0, 0, 2, 4], n)]
    ns = [int(s) for s in rng.integers(0, ocfg.frame_capacity + 1, n)]
    hard, soft, syms, stats = rx.equalize_frames(torch.from_numpy(frames).to(cuda), torch.from_numpy(taps_in).to(cuda), cn, of, ns)
    hard, soft, syms = hard.cpu().numpy(), soft.cpu().numpy(), syms.cpu().numpy()

    def same_or_close(a, b, what):
        a32, b32 = np.ascontiguousarray(a).view(np.float32), np.ascontiguousarray(b).view(np.float32)
        # the oracle's bits; NaN matches NaN; a zero matches a zero of either sign (garbage in -- zero taps, NaN carriers --
        # leaves zeros whose sign depends on how a 0 x NaN-free product was associated; on streams a transmitter produced
        # the signs agree too: tests/test_rx_gpu.py compares those bit for bit)
        ok = (a32.view(np.uint32) == b32.view(np.uint32)) | (np.isnan(a32) & np.isnan(b32)) | ((a32 == 0) & (b32 == 0))
        assert ok.all(), (what, int((~ok).sum()), a32[~ok][:4], b32[~ok][:4])

    with np.errstate(all="ignore"):
        for j in range(n):
            h_ref, s_ref, _, (snr_db, noise_db, noise_lin) = oracle.frame_equalize(ocfg, 0, cn[j], frames[j], of[j], taps_in[j])
            same_or_close(hard[j], h_ref, "hard %d" % j)
            same_or_close(soft[j], s_ref, "soft %d" % j)
            for got, want in ((stats["snr_db"][j], snr_db), (stats["noise"][j], noise_lin)):
                assert (got == want) or (np.isnan(got) and np.isnan(want)) or abs(got - want) <= 1e-5 * abs(want) + 1e-9, (j, got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(enumerate([np.inf, -np.inf, 1e-40, -1e-42, 3e38, -1e25, 1e19, 1e-39])), ids=lambda c: "%d" % c[0])
def test_block_entry_points_refuse_operands_outside_the_proven_range(oracle, dtl, cuda, case):
    """The channel estimator's and the equalisers' divisions give the oracle's bits for components that are NaN, zero or
    normal floats up to 2^62 (DTL_FUZZ_SPECIALS=1e-30,1e-25,-1e-22 passes the test above on all seeds).  Denormals,
    infinities and values from 1e20 up (squares overflow) used to give garbage that differed from the oracle's garbage;
    dtl_chanest, dtl_header_decode and dtl_equalize_frames now refuse such input (DTL_ERR_INVALID_ARG), and accept the
    same vectors once the offending component is back in range."""
    import torch
    from gr_dtl_amd import _lib
    seed, special = case
    rng = np.random.default_rng(4400 + seed)
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=64, frame_length=6)
    rx = dtl.ofdm_receiver(pcfg)
    n, N, L = 3, 64, 6

    def vec(shape):
        return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)

    def poisoned(a):
        b = a.copy().reshape(-1)
        i = int(rng.integers(0, b.size))
        b[i] = np.complex64(complex(special, b[i].imag)) if rng.integers(0, 2) else np.complex64(complex(b[i].real, special))
        return b.reshape(a.shape)
    sync, hdr, frames, taps = vec((n, 2, N)), vec((n, 2, N)), vec((n, L, N)), vec((n, N))
    off = torch.zeros(n, dtype=torch.int32, device=cuda)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    calls = {
        "chanest": lambda s, h, f, t: rx.chanest(dev(s)),
        "header_decode": lambda s, h, f, t: rx.header_decode(dev(h), dev(t), off),
        "equalize_frames": lambda s, h, f, t: rx.equalize_frames(dev(f), dev(t), [2] * n, [0] * n, [10] * n),
    }
    for name, call in calls.items():
        call(sync, hdr, frames, taps)                                   # in range: accepted
        bad_inputs = {"chanest": [(poisoned(sync), hdr, frames, taps)],
                      "header_decode": [(sync, poisoned(hdr), frames, taps), (sync, hdr, frames, poisoned(taps))],
                      "equalize_frames": [(sync, hdr, poisoned(frames), taps), (sync, hdr, frames, poisoned(taps))]}[name]
        for args in bad_inputs:
            with pytest.raises(_lib.DtlError) as e:
                call(*args)
            assert e.value.code == _lib.ERR_INVALID_ARG and "supported range" in str(e.value), (name, special)
    # NaN and zeros stay legal operands (compared with the oracle in test_random_equalizer_inputs_match_oracle)
    z = sync.copy()
    z[0, 0, 3] = np.nan
    z[1, 1, 5] = 0
    rx.chanest(dev(z))


@pytest.mark.gpu
def test_zz_fuzz_soft_values_are_the_oracles_bits():
    """Runs last in this module.  Over all receive scenarios above: taps, equalised symbols and soft bits are the oracle's
    bits, not merely within tolerance (the oscillator phase is one closed-form expression per segment and the held angle
    one fresh window sum in the same order on both sides, float math is written out on both sides); what is left are
    double-precision library calls (atan2 of the held angle, log10 / pow of the noise estimate) that land on a float
    rounding boundary."""
    from test_rx_gpu import STATS
    frac = {k: v[0] / max(v[1], 1) for k, v in STATS.items()}
    print("bit-identical fraction:", frac, {k: v[1] for k, v in STATS.items()})
    for kind, (same, total) in STATS.items():
        assert total > 100000, kind
        assert same == total, (kind, same, total)
