#!/bin/bash
# every-other-position level 1 as the default: the parity suites that reach the large-set filter, unfiltered
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04z26
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_guard.py tests/test_gpu_bench_defs.py -m gpu -x -q > gpurun_out/r04z26/pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r04z26/pytest.log
