#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02i
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -k "large_set or c4 or multi or mid_size" > "$OUT/pytest.log" 2>&1
echo "exit $?" | tee "$OUT/summary.txt"; tail -15 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
echo "== sizes" | tee -a "$OUT/summary.txt"
for mp in 1 1000000; do
  echo "ACGPU_PFX_MIN_PATTERNS=$mp" | tee -a "$OUT/summary.txt"
  ACGPU_PFX_MIN_PATTERNS=$mp timeout 600 python scripts/bench_sizes.py 2>&1 | tail -12 | tee -a "$OUT/summary.txt"
done
