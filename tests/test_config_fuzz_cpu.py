"""dtl_rx_create / dtl_tx_create against configurations no sane caller builds: every field of dtl_rx_cfg mutated at the
C boundary (negative and absurd sizes, carriers outside the transform, duplicate / empty sets, null arrays, sync words of
zeros, NaN thresholds, constellation numbers the enum does not have, truncated code tables).  The library may accept a
configuration or refuse it with a status; it may not crash, hang or abort.  The mutations run in a child process so
that a crash is a test failure with a name, not a dead test runner."""
import os
import subprocess
import sys

from conftest import ROOT

CHILD = r'''
import ctypes as C, sys, random
import numpy as np
sys.path.insert(0, %(root)r)
import gr_dtl_amd as dtl
from gr_dtl_amd import _lib
from gr_dtl_amd.ofdm_receiver import make_cfg_struct
L = _lib.lib()
L.dtl_rx_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
L.dtl_rx_destroy.argtypes = [C.c_void_p]
L.dtl_tx_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
L.dtl_tx_destroy.argtypes = [C.c_void_p]
first, last = int(sys.argv[1]), int(sys.argv[2])
INTS = ["fft_len", "cp_len", "frame_length", "has_fec", "n_occ_sets", "max_occ", "n_pilot_sets", "n_pilots",
        "n_pilot_sym_sets", "use_sync_correct", "scramble_header", "llr_dist_mode", "max_iterations", "n_fec_codes",
        "n_mcs", "initial_mcs", "decision_th", "arith", "device"]
COUNTS = {"n_occ_sets", "max_occ", "n_pilot_sets", "n_pilots", "n_pilot_sym_sets", "n_fec_codes", "n_mcs", "fft_len"}
PTRS = ["occ_len", "occ", "pilot_carriers", "pilot_symbols", "sync_word1", "sync_word2", "mcs_snr", "mcs_cnst", "mcs_fec"]
ODD_INTS = [0, -1, 1, 2, 3, 7, 48, 63, 65, 100, 4096, 1 << 20, (1 << 31) - 1, -(1 << 31)]
ODD_FLOATS = [0.0, -1.0, 1.0, 2.0, 1e30, -1e30, float("nan"), float("inf"), 1e-30]
accepted = refused = 0
for seed in range(first, last):
    rng = random.Random(seed)
    cfg = dtl.ofdm_adaptive_rx_config() if rng.random() < 0.5 else dtl.make_scaled_rx_config(rng.choice([128, 256, 1024]))
    if rng.random() < 0.7:
        cfg.fec_codes = (("fec_1", "n_0300_k_0152_gap_03.alist"),)
        cfg.mcs = ((-100000, (dtl.constellation_type_t.BPSK, "fec_1")), (11, (dtl.constellation_type_t.QPSK, "fec_1")))
        cfg.initial_mcs_id = 1
    c, keep = make_cfg_struct(cfg)
    occ_len, occ, pil, psym, sw1, sw2, arr_texts, arr_lens, mcs_snr, mcs_cnst, mcs_fec, texts = keep
    did = []
    for _ in range(rng.randint(1, 3)):
        kind = rng.randrange(8)
        did.append("kind %%d" %% kind)
        if kind == 0:
            f, v = rng.choice(INTS), rng.choice(ODD_INTS)
            if f in COUNTS:
                # a count says how far the library may read through the caller's arrays: a value between the truth and the
                # library's own limit would make it read memory the caller never had (the caller's fault, and a sanitizer's
                # finding) -- so a count is mutated DOWN, or to something the shape check must refuse
                v = rng.choice([0, -1, 1, (1 << 20), (1 << 31) - 1, -(1 << 31)])
                v = v if v <= 0 or v >= (1 << 20) else min(v, getattr(c, f))
            did[-1] = "%%s = %%d" %% (f, v)
            setattr(c, f, v)
        elif kind == 1:
            f, v = rng.choice(["sync_threshold", "eq_alpha", "bp_sigma", "hysteresis"]), rng.choice(ODD_FLOATS)
            did[-1] = "%%s = %%r" %% (f, v)
            setattr(c, f, v)
        elif kind == 2:
            f = rng.choice(PTRS)
            did[-1] = "%%s = NULL" %% f
            setattr(c, f, None)                                     # a null array
        elif kind == 3:                                             # carriers outside the transform / duplicates / DC
            a = rng.choice([occ, pil])
            a.flat[rng.randrange(a.size)] = rng.choice([0, cfg.fft_len, -cfg.fft_len, 1 << 30, -(1 << 30), int(a.flat[0])])
        elif kind == 4:
            occ_len[rng.randrange(occ_len.size)] = rng.choice([0, -5, occ.shape[1] + 1, 1 << 30])
        elif kind == 5:
            w = rng.choice([sw1, sw2]); w[:] = rng.choice([0.0, np.nan, np.inf, 1e30])
        elif kind == 6 and len(texts):
            t = texts[0][:rng.randrange(0, len(texts[0]))] + rng.choice([b"", b"\x00\xff", b" 99999999999 -3\n"])
            keep_t = C.create_string_buffer(t, len(t) + 1)
            arr_texts[0] = C.cast(keep_t, C.c_char_p); arr_lens[0] = len(t)
        elif kind == 7 and mcs_cnst.size:
            mcs_cnst[rng.randrange(mcs_cnst.size)] = rng.choice([0, 5, 6, -1, 255]); mcs_fec[rng.randrange(mcs_fec.size)] = rng.choice([0, 2, -1, 99])
    h = C.c_void_p()
    sys.stderr.write("seed %%d: %%s\n" %% (seed, "; ".join(did))); sys.stderr.flush()     # (the last line names the culprit)
    rc = L.dtl_rx_create(C.byref(c), C.byref(h))
    if rc == 0:
        accepted += 1
        assert h.value
        L.dtl_rx_destroy(h)
    else:
        refused += 1
        assert rc < 0 and not h.value, (seed, rc)
        assert L.dtl_last_error(), seed                              # a refusal says why
    ht = C.c_void_p()                                                # the transmitter takes the same structure
    rct = L.dtl_tx_create(C.byref(c), C.byref(ht))
    if rct == 0:
        L.dtl_tx_destroy(ht)
    else:
        assert rct < 0 and not ht.value, (seed, rct)
print("DONE accepted=%%d refused=%%d" %% (accepted, refused))
'''


def test_mutated_configurations_are_accepted_or_refused_never_fatal(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(CHILD % {"root": ROOT})
    total_a = total_r = 0
    for first in range(0, 1200, 300):
        out = subprocess.run([sys.executable, str(script), str(first), str(first + 300)], capture_output=True, text=True,
                             timeout=600, env=dict(os.environ, PYTHONFAULTHANDLER="1"))
        assert out.returncode == 0 and "DONE" in out.stdout, "seeds %d..%d: rc %d\n%s" % (first, first + 300, out.returncode, out.stderr[-3000:])
        a, r = [int(t.split("=")[1]) for t in out.stdout.split("DONE")[1].split()]
        total_a, total_r = total_a + a, total_r + r
    assert total_r > 300 and total_a > 50, (total_a, total_r)       # the mutations do bite, and not everything is refused
