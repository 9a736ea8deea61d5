#!/bin/bash
# round 4, GPU call F: two level-3 walks per verifier lane under the 8-byte level 1
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04f; mkdir -p $O
KEY8_VARIANTS=0,8,12 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/walk2.jsonl
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_bench_defs.py tests/test_gpu_guard.py -m gpu -x -q > $O/pytest_key8.log 2>&1; echo "key8 pytest exit $?"; tail -3 $O/pytest_key8.log
