#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02t
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee "$OUT/summary.txt"; tail -5 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
FUZZ_SECONDS=120 FUZZ_SEED=31337 timeout 400 python scripts/fuzz_gpu.py 120 31337 > "$OUT/fuzz.log" 2>&1
echo "fuzz exit $?" | tee -a "$OUT/summary.txt"; grep -v amdgpu "$OUT/fuzz.log" | tail -6 | cut -c1-400 | tee -a "$OUT/summary.txt"
