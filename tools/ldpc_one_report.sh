#!/bin/bash
# Run on the GPU box after `tools/ablate.py build 21 22 23 24 25 26` here: what a codeword costs the decoder besides its
# iterations (profiles/r04/ldpc_one_iteration.txt).  Codewords of three kinds, 2^20 each: clean (no iteration), exactly one
# iteration, and -- variant 26 -- failing the first syndrome check but not iterated; each with parts of the work removed.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$R"
echo "== clean codewords (no iteration)"
MODE=zero python3 tools/ablate.py run "python3 tools/ldpc_probe.py" base 21 22 23
echo "== codewords that need exactly one iteration"
MODE=one python3 tools/ablate.py run "python3 tools/ldpc_probe.py" base 21 22 23 24 25 26
echo "== the same in the fast arithmetic"
MODE=one ARITH=fast python3 tools/ablate.py run "python3 tools/ldpc_probe.py" base 24 26
echo "== batch sizes, one iteration each"
for L in 16 18 20 22; do MODE=one LOG2_NCW=$L python3 tools/ldpc_probe.py 2>/dev/null; done
echo "== ticket size at 2^20 (codewords per ticket; default 16)"
for C in 4 8 16 32; do echo "chunk $C: $(MODE=one DTL_LDPC_CHUNK=$C python3 tools/ldpc_probe.py 2>/dev/null)"; done
echo "== the receive chain's mix (Es/N0 2 dB on the all-zero codeword)"
MODE=mix python3 tools/ablate.py run "python3 tools/ldpc_probe.py" base 21 22 24 26
