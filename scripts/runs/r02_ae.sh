#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02ae
mkdir -p "$OUT"; : > "$OUT/summary.txt"
for lib in exp/libacgpu_pfx_10_5.so exp/libacgpu_pfx_8_8.so; do
  ACGPU_LIB=$PWD/aho-corasick_amd/lib/$lib timeout 300 python scripts/bench_c4.py 8 100000,30000,10000 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
done
