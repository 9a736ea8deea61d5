#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02l
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py -x -q -k "stream or host" > "$OUT/pytest.log" 2>&1
echo "exit $?" | tee "$OUT/summary.txt"; tail -12 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
timeout 600 python scripts/bench_host.py 2 2>&1 | grep -v amdgpu | tee -a "$OUT/summary.txt"
