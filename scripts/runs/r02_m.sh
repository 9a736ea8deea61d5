#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02m
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_enqueue.py tests/test_gpu_multi.py tests/test_c_multi.py tests/test_gpu_multirank.py -x -q > "$OUT/pytest.log" 2>&1
echo "exit $?" | tee "$OUT/summary.txt"; tail -12 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
for e in hot walk; do timeout 300 python bench.py --engine $e --no-also --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['config']['call'], 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d['roofline']['kernel'][:20])" | tee -a "$OUT/summary.txt"; done
