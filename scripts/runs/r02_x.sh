#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02x
mkdir -p "$OUT"; : > "$OUT/summary.txt"
ACGPU_PFX_MIN_PATTERNS=1 timeout 300 python scripts/bench_inputs.py --engines pf 2>/dev/null | cut -c1-60,110-270 | tee -a "$OUT/summary.txt"
