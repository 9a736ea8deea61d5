#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02k
mkdir -p "$OUT"
ACGPU_PFX_MIN_PATTERNS=1 timeout 600 python scripts/bench_inputs.py --engines pf 2>/dev/null | tee "$OUT/summary.txt"
