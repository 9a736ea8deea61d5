#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02d
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== corpora parity" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_corpora.py -x -q > "$OUT/pytest_corpora.log" 2>&1
echo "exit $?" | tee -a "$OUT/summary.txt"; tail -8 "$OUT/pytest_corpora.log" | tee -a "$OUT/summary.txt"
echo "== inputs" | tee -a "$OUT/summary.txt"
timeout 900 python scripts/bench_inputs.py > "$OUT/inputs.jsonl" 2> "$OUT/inputs.err"
echo "exit $?" | tee -a "$OUT/summary.txt"; cat "$OUT/inputs.jsonl" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/inputs.err" | tee -a "$OUT/summary.txt"
