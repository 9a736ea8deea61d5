#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02ab
mkdir -p "$OUT"; : > "$OUT/summary.txt"
for e in "" "ACGPU_PFX_NO_RING_HI=1" "ACGPU_LIB=$PWD/aho-corasick_amd/lib/exp/libacgpu_pfx_8_8.so"; do
  echo "-- $e" | tee -a "$OUT/summary.txt"
  env $e ACGPU_PFX_MIN_PATTERNS=1 timeout 300 python scripts/bench_inputs.py --engines pf --only English 2>/dev/null | grep "words-5000\|dictionary-15\|words-15000" | cut -c1-40,150-270 | tee -a "$OUT/summary.txt"
done
