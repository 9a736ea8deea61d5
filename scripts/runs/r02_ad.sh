#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02ad
mkdir -p "$OUT"; : > "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_fullsize.py -x -q -k "large_set or corpora or natural or long_prefix or c4" 2>&1 | tail -3 | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_c4.py 8 100000,30000,10000 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
ACGPU_PFX_MIN_PATTERNS=1 timeout 300 python scripts/bench_inputs.py --engines pf --only English 2>/dev/null | grep "words-5000\|dictionary-15\|words-15000\|1k random" | cut -c1-40,150-270 | tee -a "$OUT/summary.txt"
