#!/bin/bash
# Run on the GPU box after `tools/ablate.py build 21 22 23` here: everything behind
# profiles/r03/ldpc_zero_iteration.txt -- the decoder on codewords that need no iteration: time against batch size,
# against the ticket size, with parts of the work removed, and its SQ counters.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$R"
echo "== time against batch size (all codewords clean: no iteration)"
for L in 14 16 18 20 22; do LOG2_NCW=$L python3 tools/ldpc_probe.py 2>/dev/null; done
echo "== 2^20 codewords against the ticket size (codewords per ticket; default 16)"
for C in 2 4 8 16 32 64; do echo "chunk $C: $(DTL_LDPC_CHUNK=$C python3 tools/ldpc_probe.py 2>/dev/null)"; done
echo "== parts of the work removed (21 no syndrome check, 22 no output stage, 23 no soft-bit loads)"
python3 tools/ablate.py run "python3 tools/ldpc_probe.py" base 21 22 23
echo "== the receive chain's mix: Es/N0 2 dB on the all-zero codeword (about 2 iterations per codeword)"
ESN0=2 python3 tools/ldpc_probe.py 2>/dev/null
echo "== SQ counters, one launch of 2^20 clean codewords (per-launch averages)"
bash tools/ldpc_zero_pmc.sh final 2>&1 | grep -v "^tail\|^==>"
