"""GNU Radio hier block `ofdm_adaptive_rx` on the GPU block: same class name, `from_parameters` signature, ports and
message ports as the reference (python/dtl/ofdm_adaptive_rx.py:32-59), so grc/dtl_ofdm_adaptive_rx.block.yml's `make`
template works unchanged.  Needs GNU Radio 3.10 + the dtl_hip module of this directory: not importable in this repository's
test environment (gr_dtl_amd.ofdm_adaptive_rx is the GNU-Radio-free twin that the tests drive)."""
from gnuradio import gr, dtl, dtl_hip          # dtl: the reference's Python package (config dataclasses, feedback TX)

from gr_dtl_amd.ofdm_receiver import make_cfg_struct   # flattens ofdm_adaptive_rx_config into dtl_rx_cfg (ctypes)


class ofdm_adaptive_rx(gr.hier_block2):
    """Adaptive OFDM Simple modem (RX), receive graph on one MI355X."""

    @classmethod
    def from_parameters(cls, config_dict=None, **kwargs):
        cfg = dtl.ofdm_adaptive_config.make_rx_config(config_dict)
        cfg.__dict__.update(kwargs)
        return cls(cfg)

    def __init__(self, config):
        gr.hier_block2.__init__(self, "ofdm_adaptive_rx",
                                gr.io_signature(1, 1, gr.sizeof_gr_complex),
                                gr.io_signature.makev(7, 7, [gr.sizeof_char, gr.sizeof_gr_complex, gr.sizeof_char,
                                                             gr.sizeof_char, gr.sizeof_gr_complex * config.fft_len,
                                                             gr.sizeof_gr_complex, gr.sizeof_float]))
        self.message_port_register_hier_out("monitor")
        c_cfg, self._keep = make_cfg_struct(config)      # the ctypes struct IS dtl_rx_cfg; pybind takes its address
        self.direct_rx = dtl_hip.ofdm_adaptive_rx_hip(c_cfg, config.packet_length_tag_key)
        self.connect(self, self.direct_rx)
        for port in (0, 2, 3, 4, 5, 6):
            self.connect((self.direct_rx, port), (self, port))
        self.msg_connect(self.direct_rx, "monitor", self, "monitor")
        # the feedback side-channel transmitter of the reference (_setup_feedback_tx, :62-113) is unchanged host code:
        # it listens on the block's "feedback" message port and drives output port 1
        self._feedback = dtl.ofdm_adaptive_rx._setup_feedback_tx.__get__(self)
        self._feedback()
