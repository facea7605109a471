"""No C++ exception crosses the C ABI (include/dtl_rx.h).  The reference throws out of work()
(lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:148-150, lib/dtl/tb_decoder.cc:58-60); the drop-in boundary turns
those conditions into status codes, and the C++ runtime's own failures -- std::bad_alloc, std::length_error,
std::system_error -- are stopped by the handler every extern "C" body runs in (capi.cpp: guarded()).  These tests provoke
such failures on entry points that need no GPU and expect a status + error text instead of std::terminate.

The allocation failure is a real one: a child process lowers its address-space limit (RLIMIT_AS) below what the call
needs.  The child loads the library with ctypes only (no torch)."""
import ctypes as C
import os
import re
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ALIST_300, ROOT


def _sources():
    return {n: open(os.path.join(ROOT, "gr_dtl_amd", "csrc", n)).read() for n in ("capi.cpp", "rx_pipeline.cpp")}


def test_every_entry_point_body_runs_inside_the_handler():
    """Structural: each extern "C" function of capi.cpp with a body of more than one statement opens with DTL_API_BEGIN
    (one-line accessors that cannot throw are exempt), and the helper thread's job catches everything."""
    src = _sources()
    ext = src["capi.cpp"].split('extern "C" {', 1)[1]
    multi = re.findall(r"^(?:int|int64_t|void|const char\*) (dtl_\w+)\([^{;]*?\)\n\{\n(.*?)^\}\n", ext, flags=re.S | re.M)
    assert len(multi) >= 30
    for name, body in multi:
        assert body.lstrip().startswith("DTL_API_BEGIN("), name
        assert body.rstrip().endswith("DTL_API_END"), name
    one_liners = re.findall(r"^(?:int|int64_t|void|const char\*) (dtl_\w+)\([^{;]*?\) \{(.*)\}$", ext, flags=re.M)
    for name, body in one_liners:
        assert not re.search(r"std::|new |\.push_back|\.resize", body), (name, body)
    job = src["rx_pipeline.cpp"].split("auto helper_job = [&]() {", 1)[1].split("bool helper_on = false;", 1)[0]
    assert job.count("catch (...)") >= 2 and "o.ready = true" in job
    assert "std::thread helper" not in src["rx_pipeline.cpp"]          # no thread is created per call any more


def test_size_beyond_any_container_is_a_status_not_an_abort(dtl):
    L = dtl._lib.lib()
    L.dtl_fix_sync.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    t = np.zeros(4, np.int64)
    n_out = C.c_int64(0)
    # 2^59 triggers = 2^62 bytes, more than the address space: the vector's allocation throws before a byte is read
    rc = L.dtl_fix_sync(1920, t.ctypes.data, 1 << 59, 10000, t.ctypes.data, 4, C.byref(n_out))
    assert rc == dtl._lib.ERR_CAPACITY
    assert b"dtl_fix_sync" in L.dtl_last_error()
    # an alist text of 2^62 bytes: std::string(text, len) throws
    h = C.c_void_p()
    rc = L.dtl_ldpc_create_from_text(b"3 2\n", C.c_size_t(1 << 62), 15, C.c_float(0), C.byref(h))
    assert rc == dtl._lib.ERR_CAPACITY and not h.value
    # the library is still usable
    assert L.dtl_ldpc_create(os.fsencode(ALIST_300), 15, C.c_float(0), C.byref(h)) == 0
    L.dtl_ldpc_destroy(h)


_CHILD = textwrap.dedent('''
    import ctypes as C, os, resource, sys
    import numpy as np
    L = C.CDLL(sys.argv[1])
    L.dtl_last_error.restype = C.c_char_p
    L.dtl_fix_sync.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    n = 40_000_000                                    # 320 MB of trigger positions: the walk copies them (+ marks)
    trig = np.arange(n, dtype=np.int64) * 2000
    out = np.zeros(16, np.int64)
    n_out = C.c_int64(0)
    used = 0
    for line in open("/proc/self/status"):
        if line.startswith("VmSize:"):
            used = int(line.split()[1]) * 1024
    resource.setrlimit(resource.RLIMIT_AS, (used + (96 << 20), used + (96 << 20)))   # 96 MB of headroom: not enough
    rc = L.dtl_fix_sync(1920, trig.ctypes.data, n, n * 2000 + 10, out.ctypes.data, 16, C.byref(n_out))
    print("RC", rc, (L.dtl_last_error() or b"").decode())
    # ... and the process lives on: a small call works
    small = np.array([172, 2092, 4012], np.int64)
    rc2 = L.dtl_fix_sync(1920, small.ctypes.data, 3, 6000, out.ctypes.data, 16, C.byref(n_out))
    print("RC2", rc2, n_out.value)
''')


def test_allocation_failure_inside_an_entry_point_is_a_status(dtl):
    out = subprocess.run([sys.executable, "-c", _CHILD, dtl._lib.LIB_PATH], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:])     # not SIGABRT from std::terminate
    m = re.search(r"^RC (-?\d+) (.*)$", out.stdout, flags=re.M)
    assert m, out.stdout
    assert int(m.group(1)) == dtl._lib.ERR_CAPACITY, out.stdout
    assert "out of host memory" in m.group(2) or "size beyond" in m.group(2)
    m2 = re.search(r"^RC2 (-?\d+) (\d+)$", out.stdout, flags=re.M)
    assert m2 and int(m2.group(1)) == 0 and int(m2.group(2)) == 3


def test_header_symbols_must_hold_the_formatters_bit_count(dtl):
    """The reference builds its header formatter from occupied_carriers[0] alone (python/dtl/ofdm_receiver.py:129-130:
    header_len x len(occupied_carriers[0]) bits) while serializer and allocator walk the carrier sets symbol by symbol;
    with header symbols whose sets hold another number of carriers its parser and demux fall out of step.  Such a plan is
    refused when the handle is created; one whose LATER sets differ in size is accepted and reports the formatter's
    length as the row stride of the header-bit output."""
    used = [c for c in range(-26, 27) if c]
    pil = (-21, -7, 7, 21)
    occ48 = tuple(c for c in used if c not in pil)
    occ46 = occ48[1:-1]
    cfg = dtl.ofdm_adaptive_rx_config()
    cfg.fec_codes = (("fec_1", "n_0300_k_0152_gap_03.alist"),)
    cfg.mcs = ((-100000, (dtl.constellation_type_t.BPSK, "fec_1")),)
    cfg.occupied_carriers = (occ48, occ46)                      # header symbols 0, 1 -> 48 + 46 carriers, formatter: 96 bits
    import pytest
    with pytest.raises(dtl._lib.DtlError) as e:
        dtl.ofdm_receiver(cfg)
    assert e.value.code == dtl._lib.ERR_INVALID_ARG and "header" in str(e.value)
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    with pytest.raises(dtl._lib.DtlError):
        ofdm_transmitter(cfg)._device_handle()
    cfg.occupied_carriers = (occ48, occ48, occ46)               # the third set is first used by a payload symbol
    rx = dtl.ofdm_receiver(cfg)
    assert rx.frame_capacity == sum(len(cfg.occupied_carriers[i % 3]) for i in range(cfg.frame_length))
    cfg2 = dtl.ofdm_adaptive_rx_config()                        # no FEC: one header symbol, set 0
    cfg2.occupied_carriers = (occ48, occ46)
    dtl.ofdm_receiver(cfg2)
