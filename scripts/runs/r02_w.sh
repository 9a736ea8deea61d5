#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02w
mkdir -p "$OUT"; : > "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_fullsize.py -x -q -k "large_set or corpora or c4 or long_prefix" 2>&1 | tail -15 | tee -a "$OUT/summary.txt"
ACGPU_PFX_MIN_PATTERNS=1 timeout 300 python scripts/bench_inputs.py --engines pf --only English 2>/dev/null | grep "words-5000\|dictionary-15\|words-15000" | cut -c1-40,150-270 | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_inputs.py --engines auto --only English 2>/dev/null | grep "words-5000\|dictionary-15\|words-15000" | cut -c1-40,100-270 | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_c4.py 8 100000 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
