#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02h
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_c_multi.py tests/test_gpu_multi.py -x -q > "$OUT/pytest_multi.log" 2>&1
echo "exit $?" | tee "$OUT/summary.txt"; tail -25 "$OUT/pytest_multi.log" | tee -a "$OUT/summary.txt"
