"""gr_dtl_amd -- MI355X (gfx950) implementation of gr-dtl's adaptive-OFDM receive path.

Host-side mirror of the reference's operator interface (python/dtl/*.py, include/gnuradio/dtl/*.h)
over the C ABI in include/dtl_rx.h (libdtl_hip.so: hand-written HIP kernels).
"""
from . import _lib  # noqa: F401
from .fec import ldpc_dec, make_ldpc_decoders  # noqa: F401
from .ofdm_adaptive_config import (constellation_type_t, get_bits_per_symbol, make_rx_config,  # noqa: F401
                                   make_scaled_rx_config, make_tx_config, ofdm_adaptive_config,
                                   ofdm_adaptive_rx_config, ofdm_adaptive_tx_config)
from .ofdm_receiver import ofdm_adaptive_rx, ofdm_receiver  # noqa: F401
from .frame_file_store import ber_report, frame_file_store, read_frames, store_rx_frames, store_tx_frames  # noqa: F401
from .monitor import monitor_probe  # noqa: F401
from .ofdm_transmitter import ofdm_adaptive_tx, ofdm_transmitter  # noqa: F401
