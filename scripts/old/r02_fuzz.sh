#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02fuzz
mkdir -p "$OUT"
timeout 400 python scripts/fuzz_gpu.py ${FUZZ_SECONDS:-150} ${FUZZ_SEED:-20001} > "$OUT/fuzz.log" 2>&1
echo "fuzz exit $?" | tee "$OUT/summary.txt"; grep -v amdgpu "$OUT/fuzz.log" | tail -12 | cut -c1-600 | tee -a "$OUT/summary.txt"
