#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02r
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py -x -q -k "large_set or corpora" > "$OUT/pytest.log" 2>&1
echo "exit $?" | tee "$OUT/summary.txt"; tail -3 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
ACGPU_PFX_MIN_PATTERNS=1 timeout 600 python scripts/bench_inputs.py --engines pf --only English 2>/dev/null | cut -c1-270 | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_c4.py 8 100000 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_c4.py 8 100000 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
ACGPU_PFX_ONE_PASS=1 timeout 300 python scripts/bench_c4.py 8 100000 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_c4.py 8 10000 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
ACGPU_PFX_ONE_PASS=1 timeout 300 python scripts/bench_c4.py 8 10000 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
