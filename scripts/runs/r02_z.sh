#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02z
mkdir -p "$OUT"; : > "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_find.py -x -q 2>&1 | tail -3 | tee -a "$OUT/summary.txt"
timeout 300 python bench.py --no-also --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline step_ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])" | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_inputs.py --engines auto 2>/dev/null | cut -c1-70,100-270 | tee -a "$OUT/summary.txt"
