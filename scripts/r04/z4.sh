#!/bin/bash
# chain tails behind the 8-byte prefix map: A/B on natural text, then the parity suites that reach the large-set filter
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z4; mkdir -p $O
KEY8_VARIANTS=12n,12,0n,0 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/tails_ab.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_bench_defs.py tests/test_gpu_guard.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
