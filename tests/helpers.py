"""Shared builders for the RX parity tests: one scenario = product config + oracle config (same data),
a transmitted frame stream (oracle TX = test-vector generator) and AWGN."""
import os

import numpy as np

from conftest import CODES

ES = {1: 1.0, 2: 0.25, 3: 1.0, 4: 1.0}  # mean symbol energy of each constellation (dtl QPSK is scaled by 0.5)


def make_pair(oracle, dtl, fft_len=64, frame_length=20, fec="n_0300_k_0152_gap_03.alist", sync_threshold=0.95,
              use_sync_correct=True, llr_dist_mode=0, arith="gr_exact", carriers=None, scramble=False, max_it=15, bp_sigma=0.0):
    """-> (product ofdm_adaptive_rx_config, oracle Config, oracle Ldpc or None)
    carriers: (occupied sets, pilot sets, pilot symbol rows) replacing the default carrier plan."""
    if fft_len == 64:
        pcfg = dtl.ofdm_adaptive_rx_config(frame_length=frame_length)
    else:
        pcfg = dtl.make_scaled_rx_config(fft_len, frame_length=frame_length)
    if carriers is not None:
        pcfg.occupied_carriers, pcfg.pilot_carriers, pcfg.pilot_symbols = carriers
    pcfg.scramble_bits = bool(scramble)
    pcfg.ldpc_max_iterations = int(max_it)
    pcfg.ldpc_sigma = float(bp_sigma)
    pcfg.sync_threshold = sync_threshold
    pcfg.use_sync_correct = use_sync_correct
    pcfg.llr_dist_mode = llr_dist_mode
    pcfg.arith = arith
    if fec:
        pcfg.fec_codes = (("fec_1", fec),)
        pcfg.mcs = ((-100000, (dtl.constellation_type_t.BPSK, "fec_1")), (11, (dtl.constellation_type_t.QPSK, "fec_1")),
                    (16, (dtl.constellation_type_t.PSK8, "fec_1")), (21, (dtl.constellation_type_t.QAM16, "fec_1")))
        pcfg.initial_mcs_id = 1
    ocfg = oracle.Config(fft_len=fft_len, cp_len=pcfg.cp_len, frame_length=frame_length, has_fec=bool(fec),
                         sync_threshold=sync_threshold, use_sync_correct=use_sync_correct,
                         occupied_carriers=pcfg.occupied_carriers, pilot_carriers=pcfg.pilot_carriers,
                         pilot_symbols=pcfg.pilot_symbols, sync_word1=pcfg.sync_word1, sync_word2=pcfg.sync_word2,
                         llr_dist_mode=llr_dist_mode, arith={"gr_exact": 0, "fast": 1}[arith], scramble_header=bool(scramble),
                         max_it=int(max_it), bp_sigma=float(np.float32(bp_sigma)) if bp_sigma else float(np.sqrt(np.float32(2.0))))
    code = oracle.Ldpc(os.path.join(CODES, fec)) if fec else None
    return pcfg, ocfg, code


def oracle_feedback(oracle, pcfg, rx):
    return oracle.FeedbackDecision(1, 5, [(m[0], (int(m[1][0]), int(m[1][1]))) for m in rx.mcs], pcfg.initial_mcs_id)


def make_stream(oracle, ocfg, code, cnst, n_bytes, esn0_db, seed, lead_zeros=100, cnst_cycle=None):
    """TX a random payload and add AWGN at the given Es/N0 of the payload constellation."""
    rng = np.random.default_rng(seed)
    user = rng.integers(0, 256, n_bytes).astype(np.uint8)
    if code is not None:
        syms, hdrs = oracle.fec_frames_build(code, 1, cnst, ocfg.frame_capacity, user)
    else:
        syms, hdrs = oracle.nofec_frames_build(cnst, ocfg.frame_capacity, user)
    iq = oracle.tx_frames(ocfg, syms, hdrs)
    nv = 0.0 if esn0_db is None else ocfg.fft_len * ES[cnst] / 10 ** (esn0_db / 10)
    x = oracle.awgn_stream(iq, nv, seed + 1000, lead_zeros=lead_zeros)
    return user, hdrs, x
