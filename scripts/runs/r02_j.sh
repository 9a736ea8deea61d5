#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02j
mkdir -p "$OUT"
timeout 300 python scripts/bench_c4.py 8 100000,30000 2>&1 | grep patterns | tee "$OUT/summary.txt"
for so in aho-corasick_amd/lib/exp/libacgpu_pfx_*.so; do
  ACGPU_LIB=$PWD/$so timeout 300 python scripts/bench_c4.py 8 100000,30000 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
done
