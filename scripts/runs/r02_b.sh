#!/bin/bash
# round 2, second GPU call: LDS walk with 64-byte units -- parity, knob sweep, PMC of the hot kernel
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02b
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== quick parity (hot engine)" | tee "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -k "hot or misaligned or long or shards or walk" > "$OUT/pytest_quick.log" 2>&1
echo "quick exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_quick.log" | tee -a "$OUT/summary.txt"
echo "== knob sweep" | tee -a "$OUT/summary.txt"
for ch in 1 2; do for lc in 256 512 1024; do
  ACGPU_LW_CHAINS=$ch ACGPU_LW_LANE_CHUNK=$lc timeout 300 python scripts/bench_hot.py --engine hot 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
done; done
timeout 300 python scripts/bench_hot.py --engine hot --casei 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
echo "== full-size parity" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q > "$OUT/pytest_full.log" 2>&1
echo "fullsize exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_full.log" | tee -a "$OUT/summary.txt"
echo "== pmc" | tee -a "$OUT/summary.txt"
PMC_ENGINE=hot PASSES="sq1 sq2 tcc1 tcc3" BENCH_ARGS="--no-also" bash scripts/gpu_pmc.sh 2>&1 | grep -v "^$" | tail -30 | tee -a "$OUT/summary.txt"
echo "== done" | tee -a "$OUT/summary.txt"
