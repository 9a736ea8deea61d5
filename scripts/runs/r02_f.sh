#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02f
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== base (routing on)" | tee "$OUT/summary.txt"
timeout 300 python scripts/bench_hot.py --engine auto 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
ACGPU_NO_ROUTING=1 timeout 300 python scripts/bench_hot.py --engine auto 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
for so in aho-corasick_amd/lib/exp/libacgpu_exp*.so; do
  echo "== $so" | tee -a "$OUT/summary.txt"
  ACGPU_LIB=$PWD/$so timeout 300 python scripts/bench_hot.py --engine auto 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
  ACGPU_LIB=$PWD/$so timeout 300 python scripts/bench_hot.py --engine auto --alpha az --gib 1 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
done
timeout 300 python scripts/bench_hot.py --engine auto --alpha az --gib 1 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
