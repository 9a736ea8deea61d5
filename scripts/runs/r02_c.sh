#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02c
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== quick parity (hot engine)" | tee "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -k "hot or misaligned or long or shards or walk" > "$OUT/pytest_quick.log" 2>&1
echo "quick exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_quick.log" | tee -a "$OUT/summary.txt"
echo "== knob sweep" | tee -a "$OUT/summary.txt"
for un in 128 64; do for lc in 256 512 1024 2048; do
  ACGPU_LW_UNIT=$un ACGPU_LW_LANE_CHUNK=$lc timeout 300 python scripts/bench_hot.py --engine hot 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
done; done
timeout 300 python scripts/bench_hot.py --engine hot --casei 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
echo "== c5" | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_c5.py 2>&1 | tail -2 | tee -a "$OUT/summary.txt"
echo "== pmc" | tee -a "$OUT/summary.txt"
PMC_ENGINE=hot PASSES="sq2 tcc3" BENCH_ARGS="--no-also" bash scripts/gpu_pmc.sh > "$OUT/pmc.log" 2>&1
grep "lw_count" "$OUT/pmc.log" | tee -a "$OUT/summary.txt"
echo "== done" | tee -a "$OUT/summary.txt"
