#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02y
mkdir -p "$OUT"; : > "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_enqueue.py tests/test_gpu_corpora.py tests/test_gpu_multi.py -x -q -k "walk or cnfa or c4 or corpora or golden" 2>&1 | tail -4 | tee -a "$OUT/summary.txt"
timeout 600 python scripts/bench_walks.py 1 1000,10000,100000 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
