#!/usr/bin/env python3
"""Generates tests/golden/ber_tool/: frame dumps written by gr_dtl_amd.frame_file_store (the product's writer of the
reference's dump format, lib/dtl/frame_file_store.cc:30-60) and, next to each TX/RX pair, what the REFERENCE's own
tool prints for it: /root/reference/tools/ber.py is run as a program, unmodified, in the build container (it is
plain Python, no GNU Radio) and its stdout is stored as <case>.ber_py.txt.  tests/test_frame_store_cpu.py then
checks, without the reference, that (a) the writer still produces these bytes and (b) ber_report() gives the
reference tool's figures.

    python tests/golden/make_ber_fixtures.py            # needs /root/reference; rewrites the fixtures
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "ber_tool")
REF_TOOL = "/root/reference/tools/ber.py"


def scenarios():
    """name -> (tx records, rx records); a record = (frame number as carried by the 12-bit header field, payload)"""
    rng = np.random.default_rng(20240601)

    def frames(n, size, first=0):
        return [((first + i) & 0xFFF, bytes(rng.integers(0, 256, size, dtype=np.uint8))) for i in range(n)]

    out = {}
    tx = frames(40, 44)
    out["clean"] = (tx, list(tx))
    # the receiver starts late, loses frames, some frames carry bit errors
    rx = []
    for i, (n, p) in enumerate(tx):
        if i < 3 or i in (11, 12, 30):
            continue
        q = bytearray(p)
        if i in (5, 17, 18):
            q[i % len(q)] ^= 0x5A
            q[0] ^= 0x01
        rx.append((n, bytes(q)))
    out["late_start_losses_errors"] = (tx, rx)
    # frame numbers wrap at 4096 in the middle of the run; one frame comes back with another length
    tx = frames(60, 20, first=4070)
    rx = []
    for i, (n, p) in enumerate(tx):
        if i in (20, 21):
            continue
        if i == 33:
            rx.append((n, p[:-3]))
            continue
        q = bytearray(p)
        if i == 40:
            q[3] ^= 0xFF
        rx.append((n, bytes(q)))
    out["wrap_and_length_mismatch"] = (tx, rx)
    # the receive dump holds frames the transmit dump does not know (it was started later than the receiver's)
    tx_all = frames(50, 32, first=100)
    out["rx_older_than_tx"] = (tx_all[10:], [r for i, r in enumerate(tx_all) if i != 25])
    return out


def write_dump(path, records):
    sys.path.insert(0, ROOT)
    from gr_dtl_amd.frame_file_store import frame_file_store
    with frame_file_store(path) as st:
        for n, p in records:
            st.store(len(p), n, p)


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, (tx, rx) in scenarios().items():
        txf, rxf = os.path.join(OUT, name + ".tx.bin"), os.path.join(OUT, name + ".rx.bin")
        write_dump(txf, tx)
        write_dump(rxf, rx)
        res = subprocess.run([sys.executable, REF_TOOL, txf, rxf], capture_output=True, text=True, cwd=OUT)
        if res.returncode != 0:
            raise SystemExit("reference tool failed on %s: %s" % (name, res.stderr[-500:]))
        text = res.stdout.replace(OUT + os.sep, "")
        with open(os.path.join(OUT, name + ".ber_py.txt"), "w") as f:
            f.write(text)
        print(name, "->", text.strip().splitlines()[-3:])


if __name__ == "__main__":
    main()
