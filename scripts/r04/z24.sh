#!/bin/bash
# the suites the last two changes (dense event path, 4-byte ring entries) had not been run through yet
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04z24
timeout 300 python -m pytest tests/test_gpu_bench_defs.py tests/test_gpu_find.py tests/test_gpu_find_dense.py tests/test_gpu_enqueue.py tests/test_gpu_multi.py tests/test_gpu_threads.py tests/test_gpu_golden.py tests/test_gpu_stream.py tests/test_gpu_replace.py -m gpu -x -q > gpurun_out/r04z24/pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r04z24/pytest.log
