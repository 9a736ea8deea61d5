#!/bin/bash
# round 4, GPU call W: hand-scheduled piece_scan of the shallow-skip walks (DFA walk on the headline set, cNFA walk on config 4)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04w; mkdir -p $O
timeout 120 python scripts/bench_hot.py --engine walk --steps 4 2>&1 | tail -1 | tee $O/dfa_walk.json
timeout 300 python scripts/run_c4.py 8 walk 3 2>&1 | tail -1 | tee $O/c4_walk.json
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_guard.py tests/test_gpu_fullsize.py tests/test_gpu_tri_bool_flavour.py tests/test_gpu_corpora.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
