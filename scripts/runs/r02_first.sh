#!/bin/bash
# round 2, first GPU call: new LDS walk engine -- quick parity, knob sweep, full GPU suite, bench line
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02a
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== quick parity (hot engine)" | tee "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "hot or misaligned or long or shards" > "$OUT/pytest_quick.log" 2>&1
echo "quick exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_quick.log" | tee -a "$OUT/summary.txt"
echo "== knob sweep" | tee -a "$OUT/summary.txt"
for ch in 1 2; do for lc in 256 512 1024; do
  ACGPU_LW_CHAINS=$ch ACGPU_LW_LANE_CHUNK=$lc timeout 300 python scripts/bench_hot.py --engine hot 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
done; done
timeout 300 python scripts/bench_hot.py --engine hot --casei 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_hot.py --engine hot --alpha az 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_hot.py --engine walk --steps 3 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
echo "== pytest -m gpu" | tee -a "$OUT/summary.txt"
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -15 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
echo "== bench" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench exit $?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/bench.json" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/bench.err" | tee -a "$OUT/summary.txt"
echo "== done" | tee -a "$OUT/summary.txt"
