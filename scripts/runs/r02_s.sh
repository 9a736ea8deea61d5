#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02s
mkdir -p "$OUT"
: > "$OUT/summary.txt"
for cb in 680 450 300 200; do
echo "-- cb $cb" | tee -a "$OUT/summary.txt"
ACGPU_ROUTE_LS_CB=$cb timeout 600 python scripts/bench_inputs.py --engines auto --only "English" 2>/dev/null | cut -c1-270 | grep -v "1k random\|words-100\"" | tee -a "$OUT/summary.txt"
done
