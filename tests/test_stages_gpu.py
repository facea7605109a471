"""Single-stage entry points of the C ABI (dtl_sync_sc, dtl_fft_batch, dtl_soft_demap, dtl_crc32) against the oracle's
restatement of the corresponding reference blocks: integer outputs exact, floats bit-identical (same operation
sequences on both sides)."""
import zlib

import numpy as np
import pytest

from helpers import make_pair, make_stream


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 256, 1024, 128])
def test_fft_batch_matches_oracle(oracle, dtl, cuda, n):
    import torch
    from gr_dtl_amd import stages
    rng = np.random.default_rng(n)
    x = (rng.standard_normal((7, n)) + 1j * rng.standard_normal((7, n))).astype(np.complex64)
    for forward in (True, False):
        got = stages.fft_batch(torch.from_numpy(x).to(cuda), forward=forward).cpu().numpy()
        want = oracle.fft_shifted(x, forward=forward)
        np.testing.assert_array_equal(_bits(got), _bits(want))
    # forward then inverse returns N x the input (unnormalised, as FFTW / fft_vcc)
    xt = torch.from_numpy(x).to(cuda)
    back = stages.fft_batch(stages.fft_batch(xt, True), False).cpu().numpy() / n
    np.testing.assert_allclose(back, x, rtol=0, atol=2e-6 * np.sqrt(n))


@pytest.mark.gpu
@pytest.mark.parametrize("cnst", [1, 2, 3, 4])
@pytest.mark.parametrize("dist_mode", [0, 1])
def test_soft_demap_matches_oracle(oracle, dtl, cuda, cnst, dist_mode):
    import torch
    from gr_dtl_amd import stages
    rng = np.random.default_rng(10 * cnst + dist_mode)
    syms = ((rng.standard_normal(5000) + 1j * rng.standard_normal(5000)) * 0.8).astype(np.complex64)
    for noise in (None, 0.05, 0.5, 1e-6):
        got = stages.soft_demap(torch.from_numpy(syms).to(cuda), cnst, noise, dist_mode).cpu().numpy()
        want = oracle.soft_demap(cnst, syms, noise, dist_mode)
        same = (_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (cnst, dist_mode, noise, int((~same).sum()))


@pytest.mark.gpu
def test_sync_sc_matches_oracle(oracle, dtl, cuda):
    import torch
    from gr_dtl_amd import stages
    for fft_len, thr in ((64, 0.95), (256, 0.98)):
        pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=fft_len, frame_length=10, sync_threshold=thr)
        user, hdrs, x = make_stream(oracle, ocfg, code, 2, 4000, 18.0, seed=fft_len)
        trig, ang, metric = stages.sync_sc(torch.from_numpy(x).to(cuda), fft_len, ocfg.cp_len, thr, want_metric=True)
        t_ref, a_ref, m_ref = oracle.sync_sc(ocfg, x, want_metric=True)
        pos = np.nonzero(t_ref)[0]
        np.testing.assert_array_equal(trig, pos)
        assert len(trig) >= len(hdrs) - 1
        np.testing.assert_array_equal(np.asarray(ang, np.float32).view(np.uint32), np.asarray(a_ref[pos], np.float32).view(np.uint32))
        m = metric.cpu().numpy()
        ok = np.isfinite(m_ref) & np.isfinite(m)
        # window sums in f64 on both sides (different summation orders): the float metrics agree to the last bits
        np.testing.assert_allclose(m[ok], m_ref[ok], rtol=2e-6, atol=1e-30)


@pytest.mark.gpu
def test_crc32_matches_zlib(dtl, cuda):
    import torch
    from gr_dtl_amd import stages
    rng = np.random.default_rng(4)
    for n in (0, 1, 3, 63, 64, 65, 988, 4097, 1 << 20):
        data = rng.integers(0, 256, n, dtype=np.uint8)
        assert stages.crc32(torch.from_numpy(data).to(cuda)) == (zlib.crc32(data.tobytes()) & 0xFFFFFFFF), n


def test_stage_entry_points_refuse_host_memory(dtl):
    import torch
    from gr_dtl_amd import stages
    for call in (lambda: stages.fft_batch(torch.zeros(64, dtype=torch.complex64)),
                 lambda: stages.soft_demap(torch.zeros(4, dtype=torch.complex64), 2),
                 lambda: stages.crc32(torch.zeros(4, dtype=torch.uint8)),
                 lambda: stages.sync_sc(torch.zeros(100, dtype=torch.complex64), 64, 16)):
        with pytest.raises(dtl._lib.DtlError):
            call()


# ------------------------------------------------------------------------------------------------------------------
# The blocks between the FFT and the soft demapper on frequency-domain vectors (dtl_chanest, dtl_header_decode,
# dtl_equalize_frames) and the transport-block reassembly (dtl_tb_gather), each against the oracle's restatement of
# the reference block it replaces.
# ------------------------------------------------------------------------------------------------------------------
def _frames_in_frequency_domain(oracle, ocfg, code, x):
    """What fft_vcc hands the reference's header / payload paths for every frame the oracle's receiver finds:
    [(record, (2 + h + L) x N shifted vectors)]."""
    ref = oracle.rx_stream(ocfg, code, x, want=("frames", "trig", "angle", "taps", "eq_hard", "eq_soft", "soft_syms", "hdr_bits"))
    mixed = oracle.cfo_mix(ocfg, x, ref["angle"])
    N, P, cp = ocfg.fft_len, ocfg.fft_len + ocfg.cp_len, ocfg.cp_len
    n_sym = 2 + ocfg.header_len + ocfg.frame_length
    out = []
    for i, f in enumerate(ref["frames"]):
        t = f["trigger"]
        if t + n_sym * P > x.size:
            continue
        rows = np.stack([oracle.fft_shifted(mixed[t + s * P + cp:t + s * P + cp + N]) for s in range(n_sym)])
        out.append((i, f, rows.astype(np.complex64)))
    return ref, out


@pytest.mark.gpu
@pytest.mark.parametrize("fft_len,cnst,esn0,cfo", [(64, 2, 14.0, 0.0), (64, 4, 24.0, 2.0), (256, 4, 23.0, 0.0), (64, 3, 20.0, -2.3)])
def test_chanest_header_equalizer_blocks_match_oracle(oracle, dtl, cuda, fft_len, cnst, esn0, cfo):
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=fft_len, frame_length=8)
    user, hdrs, x = make_stream(oracle, ocfg, code, cnst, 6000, esn0, seed=900 + fft_len + cnst)
    if cfo:
        n = np.arange(x.size, dtype=np.float64)
        x = (x * np.exp(2j * np.pi * cfo * n / fft_len)).astype(np.complex64)
    ref, frames = _frames_in_frequency_domain(oracle, ocfg, code, x)
    assert len(frames) >= 10
    rx = dtl.ofdm_receiver(pcfg)
    h, L, N = ocfg.header_len, ocfg.frame_length, fft_len
    rows = torch.from_numpy(np.stack([r for _, _, r in frames])).to(cuda)
    # ---- ofdm_chanest_vcvc
    off, taps = rx.chanest(rows[:, 0:2])
    off_h, taps_h = off.cpu().numpy(), taps.cpu().numpy()
    for j, (i, f, r) in enumerate(frames):
        o_ref, t_ref = oracle.chanest(ocfg, r[0:2])
        assert off_h[j] == o_ref == f["carr_offset"]
        np.testing.assert_array_equal(_bits(taps_h[j]), _bits(t_ref))
    if abs(cfo) >= 1.5:
        assert (off_h != 0).all()
    # ---- header equaliser + serializer + slicer + parser
    fields, taps_out, bits = rx.header_decode(rows[:, 2:2 + h], taps, off, want_bits=True)
    bits_h, taps_out_h = bits.cpu().numpy(), taps_out.cpu().numpy()
    for j, (i, f, r) in enumerate(frames):
        assert fields["crc_ok"][j] == f["header_ok"] and fields["carr_offset"][j] == f["carr_offset"]
        np.testing.assert_array_equal(bits_h[j], ref["hdr_bits"][i])
        if f["header_ok"]:
            hd = f["hdr"]
            for k in ("payload_len", "frame_no", "fb_cnst", "tb_no", "fec_fb", "tb_offset", "fec", "tb_payload"):
                assert fields[k][j] == hd[k], (k, j)
            np.testing.assert_array_equal(_bits(taps_out_h[j]), _bits(ref["taps"][i]))
    # ---- payload frame equaliser + serializer
    ok = [j for j, (i, f, r) in enumerate(frames) if f["header_ok"]]
    assert len(ok) >= 8
    sel = torch.tensor(ok, device=cuda)
    cn = [frames[j][1]["hdr"]["cnst"] for j in ok]
    of = [frames[j][1]["carr_offset"] for j in ok]
    ns = [frames[j][1]["n_payload_syms"] for j in ok]
    hard, soft, syms, stats = rx.equalize_frames(rows[sel][:, 2 + h:], taps_out[sel], cn, of, ns)
    hard, soft, syms = hard.cpu().numpy(), soft.cpu().numpy(), syms.cpu().numpy()
    for q, j in enumerate(ok):
        i, f, r = frames[j]
        np.testing.assert_array_equal(hard[q], ref["eq_hard"][i])
        same = _bits(soft[q]) == _bits(ref["eq_soft"][i])
        assert same.all()
        nsym = min(f["n_payload_syms"], ocfg.frame_capacity) if f["hdr"]["payload_len"] else 0
        np.testing.assert_array_equal(_bits(syms[q][:nsym]), _bits(ref["soft_syms"][i][:nsym]))
        assert abs(stats["snr_db"][q] - f["snr_db"]) <= 1e-9               # (f64 Welford merge order: tests/test_rx_gpu.py)
        assert abs(stats["noise"][q] - f["noise_lin"]) <= 1e-10 * abs(f["noise_lin"])
    # a frame without a valid constellation is the reference's "Missing constellation tag." exception
    with pytest.raises(dtl._lib.DtlError) as e:
        rx.equalize_frames(rows[sel][:1, 2 + h:], taps_out[sel][:1], [0], [0], [10])
    assert e.value.code == dtl._lib.ERR_TAGS_MISSING


@pytest.mark.gpu
@pytest.mark.parametrize("cnst,esn0,frame_length", [(2, 13.0, 20), (4, 23.0, 20), (1, 20.0, 3), (3, 21.0, 5)])
def test_tb_gather_matches_receiver_and_oracle(oracle, dtl, cuda, cnst, esn0, frame_length):
    """dtl_tb_gather + dtl_ldpc_decode_gather on the soft bits of a receiver run reproduce the run's own transport
    blocks (= the oracle's tb_decoder, test_rx_gpu): block numbers, codeword counts, CRC flags, payload bytes.
    frame_length 3 / 5: blocks spread over several short frames (more than four pieces before merging)."""
    import ctypes as C
    import torch
    import zlib
    from conftest import ALIST_300
    pcfg, ocfg, code = make_pair(oracle, dtl, frame_length=frame_length)
    user, hdrs, x = make_stream(oracle, ocfg, code, cnst, 5000, esn0, seed=40 + cnst)
    rx = dtl.ofdm_receiver(pcfg)
    res = rx.work(torch.from_numpy(x).to(cuda), want=("llr",))
    ref = oracle.rx_stream(ocfg, code, x)
    assert [t["tb_no"] for t in res.tbs] == [t["tb_no"] for t in ref["tbs"]] and len(res.tbs) >= 5
    desc, spans = rx.tb_gather(res.frames, [(code.n, code.k)], cuda)
    assert [int(s) for s in spans["tb_no"]] == [int(t["tb_no"]) for t in res.tbs]
    assert [int(s) for s in spans["tb_payload_bits"]] == [int(t["tb_payload"]) for t in res.tbs]
    n_cw = desc.numel() // C.sizeof(dtl._lib.CwDesc)
    assert n_cw == int(spans["n_cw"].sum()) == res.n_codewords
    dec = dtl.ldpc_dec(ALIST_300)
    bits = torch.zeros(int(spans["bits_off"][-1]) + int(spans["tb_payload_bits"][-1]) + 64, dtype=torch.uint8, device=cuda)
    nit = torch.zeros(n_cw, dtype=torch.int32, device=cuda)
    dec.decode_gather(res.llr, desc, bits, nit)
    bits = bits.cpu().numpy()
    payload = res.payload.cpu().numpy()
    for sp, t in zip(spans, res.tbs):
        b = bits[sp["bits_off"]:sp["bits_off"] + sp["tb_payload_bits"]]
        by = np.packbits(b.reshape(-1, 8)[:, ::-1], axis=1).ravel()
        crc_ok = int(by.size >= 4 and (zlib.crc32(by[:-4].tobytes()) & 0xFFFFFFFF) == int.from_bytes(by[-4:].tobytes(), "little"))
        assert crc_ok == int(t["crc_ok"])
        if crc_ok:
            np.testing.assert_array_equal(by[:-4], payload[t["out_offset"]:t["out_offset"] + t["user_bits"] // 8])
    assert sum(int(t["crc_ok"]) for t in res.tbs) >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("cnst", [2, 3, 4])
def test_equalizer_with_pilot_sets_of_different_sizes(oracle, dtl, cuda, cnst):
    """The carrier plan of the reference's own equaliser QA (qa_ofdm_adaptive_frame_equalizer_vcvc.py:80-85) has pilot
    sets of different sizes -- three empty ones and one that takes every carrier -- which the flattened C-ABI tables pad
    with DTL_NO_CARRIER.  Same structure at 64 carriers: dtl_equalize_frames against the oracle's frame equaliser."""
    import torch
    rng = np.random.default_rng(60 + cnst)
    pcfg = dtl.ofdm_adaptive_rx_config(frame_length=6)
    occ = tuple(pcfg.occupied_carriers[0])
    pil = tuple(pcfg.pilot_carriers[0])
    pcfg.pilot_carriers = (pil, (), pil[:2], occ[:8] + pil)          # 4, 0, 2 and 12 pilots
    pcfg.pilot_symbols = tuple(tuple(complex(v) for v in rng.choice([-1.0, 1.0], len(p))) for p in pcfg.pilot_carriers)
    pcfg.fec_codes = (("fec_1", "n_0300_k_0152_gap_03.alist"),)
    pcfg.mcs = ((-100000, (dtl.constellation_type_t.BPSK, "fec_1")), (11, (dtl.constellation_type_t.QPSK, "fec_1")))
    ocfg = oracle.Config(fft_len=64, cp_len=16, frame_length=6, has_fec=True, occupied_carriers=pcfg.occupied_carriers,
                         pilot_carriers=pcfg.pilot_carriers, pilot_symbols=pcfg.pilot_symbols,
                         sync_word1=pcfg.sync_word1, sync_word2=pcfg.sync_word2)
    rx = dtl.ofdm_receiver(pcfg)
    n, L, N = 12, 6, 64
    pts = oracle.constellation_points(cnst)
    frames = np.zeros((n, L, N), np.complex64)
    taps = np.zeros((n, N), np.complex64)
    shifted = lambda c: ((c + N) % N + N // 2) % N
    for f in range(n):
        h = np.exp(1j * rng.uniform(-0.4, 0.4, N)) * rng.uniform(0.8, 1.2, N)
        taps[f] = h.astype(np.complex64)
        for i in range(L):
            pset = pcfg.pilot_carriers[(2 + i) % 4]                 # symbols_skipped = header_len = 2
            psym = pcfg.pilot_symbols[(2 + i) % 4]
            for c in occ:
                frames[f, i, shifted(c)] = pts[rng.integers(0, len(pts))]
            for c, v in zip(pset, psym):
                frames[f, i, shifted(c)] = v
            drift = np.exp(1j * 0.02 * (i + 1))
            frames[f, i] = frames[f, i] * h * drift + (frames[f, i] != 0) * (rng.normal(0, 0.02, N) + 1j * rng.normal(0, 0.02, N))
    hard, soft, syms, stats = rx.equalize_frames(torch.from_numpy(frames).to(cuda), torch.from_numpy(taps).to(cuda),
                                                 [cnst] * n, [0] * n, [rx.frame_capacity] * n)
    hard, soft = hard.cpu().numpy(), soft.cpu().numpy()
    for f in range(n):
        h_ref, s_ref, t_ref, (snr_db, noise_db, noise_lin) = oracle.frame_equalize(ocfg, False, cnst, frames[f], 0, taps[f])
        np.testing.assert_array_equal(hard[f], h_ref.reshape(L, N))
        np.testing.assert_array_equal(_bits(soft[f]), _bits(s_ref.reshape(L, N)))
        assert abs(stats["snr_db"][f] - snr_db) <= 1e-9


@pytest.mark.gpu
def test_block_entry_points_with_absurd_carrier_offsets(oracle, dtl, cuda):
    """The carrier offset handed to dtl_header_decode / dtl_equalize_frames is the caller's (a tag in the reference): at or
    beyond +-(symbols x fft_len) every shifted position falls outside the block, so every larger value -- up to the ends of
    the int range, where a sum or a negation would overflow -- must give what that one gives, and must not fault."""
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl, frame_length=8)
    user, hdrs, x = make_stream(oracle, ocfg, code, 2, 3000, 20.0, seed=77)
    ref, frames = _frames_in_frequency_domain(oracle, ocfg, code, x)
    rx = dtl.ofdm_receiver(pcfg)
    h, L, N = ocfg.header_len, ocfg.frame_length, ocfg.fft_len
    rows = torch.from_numpy(np.stack([r for _, _, r in frames[:6]])).to(cuda)
    off, taps = rx.chanest(rows[:, 0:2])
    imin, imax = -(1 << 31), (1 << 31) - 1
    # ---- header: +-h N against the ends of the int range (and a value in between)
    outs = {}
    for v in (h * N, imax, 1 << 30, -h * N, imin, -(1 << 30)):
        o = torch.full_like(off, v)
        fields, taps_out, bits = rx.header_decode(rows[:, 2:2 + h], taps, o, want_bits=True)
        outs[v] = (bits.cpu().numpy(), taps_out.cpu().numpy())
        assert not fields["crc_ok"].any()
    for big, edge in ((imax, h * N), (1 << 30, h * N), (imin, -h * N), (-(1 << 30), -h * N)):
        np.testing.assert_array_equal(outs[big][0], outs[edge][0])
    # ---- payload: the same at +-L N
    fields, taps_out = rx.header_decode(rows[:, 2:2 + h], taps, off)
    n = rows.shape[0]
    res = {}
    for v in (L * N, imax, -L * N, imin, 12345678):
        hard, soft, syms, stats = rx.equalize_frames(rows[:, 2 + h:], taps_out, [2] * n, [v] * n, [ocfg.frame_capacity] * n)
        res[v] = (hard.cpu().numpy(), syms.cpu().numpy())
    for big, edge in ((imax, L * N), (12345678, L * N), (imin, -L * N)):
        np.testing.assert_array_equal(_bits(res[big][0]), _bits(res[edge][0]))
        np.testing.assert_array_equal(_bits(res[big][1]), _bits(res[edge][1]))
    # and an ordinary call afterwards is the ordinary result
    hard, soft, syms, stats = rx.equalize_frames(rows[:, 2 + h:], taps_out, [2] * n, [int(v) for v in off.cpu().numpy()], [ocfg.frame_capacity] * n)
    for q in range(n):
        i, f, r = frames[q]
        if f["header_ok"]:
            np.testing.assert_array_equal(hard.cpu().numpy()[q], ref["eq_hard"][i])
