#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02u
mkdir -p "$OUT"; : > "$OUT/summary.txt"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py -x -q -k "large_set or corpora" 2>&1 | tail -2 | tee -a "$OUT/summary.txt"
for lib in libacgpu.so exp/libacgpu_pfx_8_8.so; do
  echo "-- $lib" | tee -a "$OUT/summary.txt"
  ACGPU_LIB=$PWD/aho-corasick_amd/lib/$lib ACGPU_PFX_MIN_PATTERNS=1 timeout 300 python scripts/bench_inputs.py --engines pf --only English 2>/dev/null | grep "words-5000\|dictionary-15\|words-15000" | cut -c1-40,150-270 | tee -a "$OUT/summary.txt"
  ACGPU_LIB=$PWD/aho-corasick_amd/lib/$lib timeout 300 python scripts/bench_c4.py 8 100000 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
done
