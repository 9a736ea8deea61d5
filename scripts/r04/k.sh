#!/bin/bash
# round 4, GPU call K: LDS walk with odd row strides (bank spread) vs the power-of-two rows, ascii and a-z; parity
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04k; mkdir -p $O
timeout 100 python scripts/bench_hot.py --steps 8 2>&1 | tail -1 | tee $O/hot_ascii_odd.json
ACGPU_LW_POW2_ROWS=1 timeout 100 python scripts/bench_hot.py --steps 8 2>&1 | tail -1 | tee $O/hot_ascii_pow2.json
timeout 100 python scripts/bench_hot.py --steps 8 --alpha az 2>&1 | tail -1 | tee $O/hot_az_odd.json
ACGPU_LW_POW2_ROWS=1 timeout 100 python scripts/bench_hot.py --steps 8 --alpha az 2>&1 | tail -1 | tee $O/hot_az_pow2.json
timeout 100 python scripts/bench_hot.py --steps 8 --casei 2>&1 | tail -1 | tee $O/hot_casei_odd.json
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_guard.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
