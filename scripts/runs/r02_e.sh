#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02e
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee "$OUT/summary.txt"
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -15 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
echo "== inputs" | tee -a "$OUT/summary.txt"
timeout 900 python scripts/bench_inputs.py --engines auto,pf,hot > "$OUT/inputs.jsonl" 2> "$OUT/inputs.err"
echo "exit $?" | tee -a "$OUT/summary.txt"; cat "$OUT/inputs.jsonl" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/inputs.err" | tee -a "$OUT/summary.txt"
echo "== bench" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench exit $?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/bench.json" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/bench.err" | tee -a "$OUT/summary.txt"
