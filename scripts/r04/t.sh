#!/bin/bash
# round 4, GPU call T: half-size table + 512-entry rings for the 8-byte level 1, A/B against the 128 KiB table; parity
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04t; mkdir -p $O
KEY8_VARIANTS=12 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/small.jsonl
ACGPU_PFX_KEY8_BIG_TABLE=1 KEY8_VARIANTS=12 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/big.jsonl
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_bench_defs.py tests/test_gpu_guard.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
