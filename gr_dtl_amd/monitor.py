"""Monitor counters of the receive path, in the shape of the reference's telemetry messages
(lib/dtl/proto/monitor_ofdm.proto:3-22): one `monitor_eq_msg` per decoded frame
(ofdm_adaptive_frame_equalizer_vcvc_impl.cc:196-216) and one `monitor_dec_msg` per transport block
(ofdm_adaptive_fec_decoder_impl.cc:182-195).  The reference publishes them on message ports as protobuf
blobs; here they are plain dicts built from the record arrays of an `rx_result`, with the running counters
(CRC ok / fail, lost frames) carried across work() / feed() calls by a `monitor_probe`."""
from .ofdm_adaptive_config import get_bits_per_symbol


class monitor_probe:
    def __init__(self):
        self.crc_ok_count = 0
        self.crc_fail_count = 0
        self.frames_count = 0      # d_frames_count: decoded + lost
        self.lost_frames = 0       # d_lost_frames

    def eq_msgs(self, rx_result):
        """monitor_eq_msg per payload frame.  `lost_frames` in the frame records is cumulative over the stream."""
        out = []
        for f in rx_result.frames:
            if f["payload_index"] < 0:
                continue
            lost_now = int(f["lost_frames"]) - self.lost_frames if int(f["lost_frames"]) >= self.lost_frames \
                else int(f["lost_frames"])            # a new stream restarts the counter
            self.lost_frames += lost_now
            self.frames_count += lost_now + 1
            out.append(dict(constellation_key=int(f["feedback_cnst"]), fec_key=int(f["feedback_fec"]),
                            estimated_snr_tag_key=float(f["snr_db"]), noise_tag_key=float(f["noise"]),
                            lost_frames_rate=100.0 * self.lost_frames / self.frames_count))
        return out

    def dec_msgs(self, rx_result, code_n=0, code_k=0):
        """monitor_dec_msg per transport block (FEC chain) or per packed frame (no-FEC chain)."""
        bps_by_tb = {}
        payload_by_tb = {}
        for f in rx_result.frames:
            if f["payload_index"] >= 0:
                bps_by_tb.setdefault(int(f["tb_no"]), get_bits_per_symbol(int(f["cnst"])))
                payload_by_tb[int(f["tb_no"])] = int(f["payload_len"])
        out = []
        for t in rx_result.tbs:
            if t["crc_ok"]:
                self.crc_ok_count += 1
            else:
                self.crc_fail_count += 1
            total = self.crc_ok_count + self.crc_fail_count
            n_cw = 0
            if code_n:
                frame_len = 8 * payload_by_tb.get(int(t["tb_no"]), 0)
                n_cw = 1 + frame_len // code_n if frame_len > code_n else 1
            out.append(dict(tb_no=int(t["tb_no"]), tb_payload=int(t["tb_payload"]), tb_code_k=int(code_k),
                            tb_code_n=int(code_n), tb_codewords=n_cw,
                            frame_payload=payload_by_tb.get(int(t["tb_no"]), 0),
                            bps=bps_by_tb.get(int(t["tb_no"]), 0),
                            crc_ok_count=self.crc_ok_count, crc_fail_count=self.crc_fail_count,
                            tber=100.0 * self.crc_fail_count / total, avg_it=float(t["avg_it"])))
        return out
