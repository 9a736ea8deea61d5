#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02ac
mkdir -p "$OUT"; : > "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_fullsize.py tests/test_gpu_guard.py -x -q -k "large_set or corpora or natural or long_prefix or guard" 2>&1 | tail -3 | tee -a "$OUT/summary.txt"
timeout 400 python scripts/fuzz_gpu.py 150 52000 2>&1 | grep -v amdgpu | tail -4 | cut -c1-500 | tee -a "$OUT/summary.txt"
