"""gr_dtl_amd -- MI355X (gfx950) implementation of gr-dtl's adaptive-OFDM receive path.

Host-side mirror of the reference's operator interface (python/dtl/*.py, include/gnuradio/dtl/*.h)
over the C ABI in include/dtl_rx.h (libdtl_hip.so: hand-written HIP kernels).
"""
from . import _lib  # noqa: F401
from .fec import ldpc_dec, make_ldpc_decoders  # noqa: F401
