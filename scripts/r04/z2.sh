#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z2; mkdir -p $O
timeout 200 python scripts/bench_nat.py 40 2>&1 | grep '^{' | cut -c1-330 | tee $O/nat.jsonl
timeout 600 python -m pytest tests/test_gpu_corpora.py tests/test_gpu_enqueue.py tests/test_gpu_parity.py tests/test_gpu_bench_defs.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
