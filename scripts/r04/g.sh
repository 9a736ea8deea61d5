#!/bin/bash
# round 4, GPU call G: rolled verifier loops (smaller kernels) -- natural text roles, config 4 regression check, parity
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04g; mkdir -p $O
KEY8_VARIANTS=0,8,12,14 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/rolled.jsonl
timeout 200 python scripts/run_c4.py 8 auto 5 2>&1 | tail -1 | tee $O/c4.json
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_bench_defs.py tests/test_gpu_guard.py -m gpu -x -q > $O/pytest_key8.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_key8.log
