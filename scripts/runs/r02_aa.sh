#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02aa
mkdir -p "$OUT"; : > "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_corpora.py tests/test_gpu_fullsize.py tests/test_gpu_guard.py -x -q -k "hot or golden or corpora or c2 or full or guard" 2>&1 | tail -3 | tee -a "$OUT/summary.txt"
for e in "" "ACGPU_LW_NO_AFFINE=1"; do
  echo "-- $e" | tee -a "$OUT/summary.txt"
  env $e timeout 300 python bench.py --engine hot --no-also --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hot step_ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])" | tee -a "$OUT/summary.txt"
done
timeout 300 python scripts/bench_inputs.py --engines hot 2>/dev/null | cut -c1-70,100-270 | tee -a "$OUT/summary.txt"
