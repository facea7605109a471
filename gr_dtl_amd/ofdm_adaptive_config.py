"""Configuration dataclasses of the adaptive-OFDM PHY.

Mirror of python/dtl/ofdm_adaptive_config.py:10-106 (same field names and defaults, same JSON
overrides through make_rx_config / make_tx_config) without the GNU Radio imports: the sync words
that the reference obtains from gnuradio.digital.ofdm_txrx._make_sync_word1/2 are rebuilt here with
the same seeded legacy numpy generator.
"""
import dataclasses as dc
import enum
import sys
import typing as t

import numpy as np


class constellation_type_t(enum.IntEnum):
    """include/gnuradio/dtl/ofdm_adaptive_utils.h:22-28"""
    UNKNOWN = 0
    BPSK = 1
    QPSK = 2
    PSK8 = 3
    QAM16 = 4


_BITS_PER_SYMBOL = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4}  # lib/dtl/ofdm_adaptive_utils.cc:64-70


def get_bits_per_symbol(constellation):
    return _BITS_PER_SYMBOL.get(int(constellation), 0)


_SEQ_SEED = 42  # gnuradio.digital.ofdm_txrx._seq_seed


def _get_active_carriers(fft_len, occupied_carriers, pilot_carriers):
    active = []
    for carrier in list(occupied_carriers[0]) + list(pilot_carriers[0]):
        if carrier < 0:
            carrier += fft_len
        active.append(carrier)
    return set(active)


def _make_sync_word1(fft_len, occupied_carriers, pilot_carriers):
    """BPSK * sqrt(2) on the odd active carriers, zero elsewhere (two identical halves in time)."""
    active = _get_active_carriers(fft_len, occupied_carriers, pilot_carriers)
    rs = np.random.RandomState(_SEQ_SEED)
    bpsk = {0: np.sqrt(2), 1: -np.sqrt(2)}
    sw1 = [bpsk[rs.randint(2)] if (x in active and x % 2) else 0 for x in range(fft_len)]
    return np.fft.fftshift(sw1)


def _make_sync_word2(fft_len, occupied_carriers, pilot_carriers):
    """BPSK on every active carrier, DC zero."""
    active = _get_active_carriers(fft_len, occupied_carriers, pilot_carriers)
    rs = np.random.RandomState(_SEQ_SEED)
    bpsk = {0: 1, 1: -1}
    sw2 = [bpsk[rs.randint(2)] if x in active else 0 for x in range(fft_len)]
    sw2[0] = 0j
    return np.fft.fftshift(sw2)


_PILOT_SYM_SCRAMBLE_SEQ = (
    1, 1, 1, 1, -1, -1, -1, 1, -1, -1, -1, -1, 1, 1, -1, 1, -1, -1, 1, 1, -1, 1, 1, -1, 1, 1, 1, 1, 1, 1, -1, 1,
    1, 1, -1, 1, 1, -1, -1, 1, 1, 1, -1, 1, -1, -1, -1, 1, -1, 1, -1, -1, 1, -1, -1, 1, 1, 1, 1, 1, -1, -1, 1, 1,
    -1, -1, 1, -1, 1, -1, 1, 1, -1, -1, -1, 1, 1, -1, -1, -1, -1, 1, -1, -1, 1, -1, 1, 1, 1, 1, -1, 1, -1, 1, -1, 1,
    -1, -1, -1, -1, -1, 1, -1, 1, 1, -1, 1, -1, 1, 1, 1, -1, -1, 1, -1, -1, -1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1)

_OCCUPIED_64 = (list(range(-26, -21)) + list(range(-20, -7)) + list(range(-6, 0)) + list(range(1, 7))
                + list(range(8, 21)) + list(range(22, 27)),)
_PILOTS_64 = ((-21, -7, 7, 21,), (-21, -7, 7, 21,))


def _mcs_default():
    return ((sys.float_info.min, (constellation_type_t.BPSK, "no_fec")),
            (13, (constellation_type_t.QPSK, "no_fec")),
            (18, (constellation_type_t.PSK8, "no_fec")),
            (23, (constellation_type_t.QAM16, "no_fec")),)


@dc.dataclass
class ofdm_adaptive_config:
    fft_len: int = 64
    cp_len: int = 16
    frame_length_tag_key: str = "frame_length"
    packet_length_tag_key: str = "packet_length"
    frame_no_tag_key: str = "frame_no"
    occupied_carriers: t.Tuple[t.List[int]] = _OCCUPIED_64
    pilot_carriers: t.Tuple[t.Tuple[int]] = _PILOTS_64
    pilot_sym_scramble_seq: t.Tuple[int] = _PILOT_SYM_SCRAMBLE_SEQ
    pilot_symbols: t.Tuple[t.Tuple[int]] = tuple((x, x, x, -x) for x in _PILOT_SYM_SCRAMBLE_SEQ)
    sync_word1: t.Any = None
    sync_word2: t.Any = None
    rolloff: int = 0
    scramble_bits: bool = False
    frame_length: int = 20
    frame_store_folder: str = "/tmp"
    fec: bool = False
    fec_codes: t.Tuple[t.Tuple[str, str]] = ()
    mcs: t.Tuple[t.Tuple[float, t.Tuple[constellation_type_t, str]]] = dc.field(default_factory=_mcs_default)
    initial_mcs_id: int = 0

    def __post_init__(self):
        if self.sync_word1 is None:
            self.sync_word1 = _make_sync_word1(self.fft_len, self.occupied_carriers, self.pilot_carriers)
        if self.sync_word2 is None:
            self.sync_word2 = _make_sync_word2(self.fft_len, self.occupied_carriers, self.pilot_carriers)


@dc.dataclass
class ofdm_adaptive_tx_config(ofdm_adaptive_config):
    max_empty_frames: int = -1
    sample_rate: int = 700000


@dc.dataclass
class ofdm_adaptive_rx_config(ofdm_adaptive_config):
    sync_threshold: float = 0.95
    use_sync_correct: bool = True
    # extensions of this implementation (not in the reference dataclass)
    llr_dist_mode: int = 0      # constellation::calc_soft_dec distance: 0 squared, 1 plain
    ldpc_sigma: float = 0.0     # awgn_bp sigma (never set by gr-dtl); 0 -> sqrt(2)
    ldpc_max_iterations: int = 15
    arith: str = "gr_exact"     # "gr_exact": the reference's operations (its bits); "fast": production arithmetic
                                # of the BP decoder and the 16-QAM demapper (include/dtl_rx.h, enum dtl_arith)
    device: int = -1            # dtl_rx_cfg.device: the GPU this receiver's handle is bound to; -1 = whichever device is
                                # current when a call is made (one process per GPU: the usual arrangement)


@dc.dataclass
class ofdm_adaptive_full_duplex_config(ofdm_adaptive_config):
    sync_threshold: float = 0.95
    use_sync_correct: bool = True
    max_empty_frames: int = -1
    sample_rate: int = 700000


def _make_config(cfg, json_dict, parser):
    """python/dtl/ofdm_adaptive_config.py:66-89"""
    class _default_parser:
        def mcs(cls, v):
            cnsts = {"bpsk": constellation_type_t.BPSK, "qpsk": constellation_type_t.QPSK,
                     "psk8": constellation_type_t.PSK8, "qam16": constellation_type_t.QAM16}
            return [(snr, (cnsts[cnst], fec)) for (snr, (cnst, fec)) in v]

    if json_dict:
        for key, val in json_dict.items():
            if key in cfg.__dict__:
                if key in parser.__dict__:
                    cfg.__setattr__(key, parser.__dict__[key](parser, val))
                elif key in _default_parser.__dict__:
                    cfg.__setattr__(key, _default_parser.__dict__[key](_default_parser, val))
                else:
                    cfg.__setattr__(key, val)
    return cfg


def make_tx_config(json_dict):
    class _parser:
        pass
    return _make_config(ofdm_adaptive_tx_config(), json_dict, _parser)


def make_rx_config(json_dict):
    class _parser:
        pass
    return _make_config(ofdm_adaptive_rx_config(), json_dict, _parser)


def make_full_duplex_config(json_dict):
    class _parser:
        pass
    return _make_config(ofdm_adaptive_full_duplex_config(), json_dict, _parser)


def scaled_carrier_map(fft_len):
    """Carrier map for fft_len = 64*R (the reference ships only the 64-carrier map): every base
    carrier b of the 64-map becomes the R carriers b*R .. b*R +/- (R-1) (192 data +
    16 pilots at 256).  Returns (occupied_carriers, pilot_carriers, pilot_symbols)."""
    R = fft_len // 64
    if fft_len != 64 * R or R < 1:
        raise ValueError("fft_len must be a multiple of 64")

    def widen(b):
        return [b * R + j if b > 0 else b * R - j for j in range(R)]
    occ = sorted(c for b in _OCCUPIED_64[0] for c in widen(b))
    pil = sorted(c for b in _PILOTS_64[0] for c in widen(b))
    sign = {b: s for b, s in zip(_PILOTS_64[0], (1, 1, 1, -1))}
    pil_sign = [sign[int(np.sign(c) * (abs(c) // R))] for c in pil]
    pilot_symbols = tuple(tuple(x * s for s in pil_sign) for x in _PILOT_SYM_SCRAMBLE_SEQ)
    return (occ,), (tuple(pil), tuple(pil)), pilot_symbols


def make_scaled_rx_config(fft_len, **kwargs):
    occ, pil, psym = scaled_carrier_map(fft_len)
    base = dict(fft_len=fft_len, cp_len=fft_len // 4, occupied_carriers=occ, pilot_carriers=pil,
                pilot_symbols=psym)
    base.update(kwargs)
    return ofdm_adaptive_rx_config(**base)
