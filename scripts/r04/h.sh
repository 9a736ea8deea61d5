#!/bin/bash
# round 4, GPU call H: leftmost find_iter from the per-start table -- parity tests, then the dense reference definitions
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_find_dense.py -m gpu -x -q > $O/pytest_dense.log 2>&1; echo "dense pytest exit $?"; tail -15 $O/pytest_dense.log
timeout 300 python -m pytest tests/test_gpu_find.py tests/test_gpu_bench_defs.py -m gpu -x -q > $O/pytest_find.log 2>&1; echo "find pytest exit $?"; tail -3 $O/pytest_find.log
timeout 300 python scripts/bench_defs.py 256 auto onebyte-match,teddy1-1pat-common,teddy1-16pat-common,big16earlyshort,teddy3-64pat-common,sorted.txt 2>&1 | grep '"bench"' | cut -c1-400 | tee $O/defs.jsonl
