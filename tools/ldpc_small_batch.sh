#!/bin/bash
# Run on the GPU box: the decoder on small batches (2^15 codewords: the C2 step's size) -- what a launch costs when the
# batch does not fill the machine for long: clean codewords (fixed cost), the receive chain's mix at 1 dB (a few percent
# of the codewords at the iteration limit, like C2 at 12 dB), ticket sizes and ticket stealing.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$R"
for L in 12 15; do
  echo "== 2^$L codewords"
  MODE=zero LOG2_NCW=$L python3 tools/ldpc_probe.py 2>/dev/null
  for E in 3 1 0; do
    echo "-- mix at $E dB"
    for C in 1 2 4; do echo "chunk $C steal off: $(MODE=mix ESN0=$E LOG2_NCW=$L DTL_LDPC_CHUNK=$C python3 tools/ldpc_probe.py 2>/dev/null)"; done
    for C in 1 2; do echo "chunk $C steal on : $(MODE=mix ESN0=$E LOG2_NCW=$L DTL_LDPC_CHUNK=$C DTL_LDPC_STEAL_MIN=1 python3 tools/ldpc_probe.py 2>/dev/null)"; done
  done
done
