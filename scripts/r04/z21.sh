#!/bin/bash
# one host round trip less on the dense event path (find_iter's occurrence stream, config 5): parity of every caller, c5 timing
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z21; mkdir -p $O
timeout 120 python scripts/bench_c5.py 2>&1 | tail -2 | cut -c1-260 | tee $O/c5.jsonl
timeout 600 python -m pytest tests/test_gpu_find.py tests/test_gpu_find_dense.py tests/test_gpu_stream.py tests/test_gpu_replace.py tests/test_gpu_golden.py tests/test_gpu_bench_defs.py tests/test_gpu_corpora.py tests/test_gpu_enqueue.py tests/test_gpu_threads.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
timeout 120 python scripts/bench_c5.py 2>&1 | tail -2 | cut -c1-260 | tee -a $O/c5.jsonl
