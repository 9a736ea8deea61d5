#!/bin/bash
# One gpurun call: GPU tests -> smoke -> bench (engines) -> rocprofv3 kernel-trace summary.
# Everything is logged under gpurun_out/ (merged back by gpurun).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/$(date +%H%M%S)
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== env" | tee "$OUT/summary.txt"
(rocm-smi --showproductname 2>/dev/null | head -8; nproc; free -g | head -2) >> "$OUT/summary.txt" 2>&1

if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== pytest -m gpu" | tee -a "$OUT/summary.txt"
  timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" | tee -a "$OUT/summary.txt"
  tail -15 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
  echo "== smoke" | tee -a "$OUT/summary.txt"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
  echo "smoke exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
fi

echo "== bench" | tee -a "$OUT/summary.txt"
SPECS=${BENCH_SPECS-hot:8:0 walk:8:0}   # BENCH_SPECS="" skips the engine benches
for spec in $SPECS; do
  IFS=: read -r eng gib chunk <<< "$spec"
  timeout 900 python bench.py --engine "$eng" --gib "$gib" --chunk "$chunk" --steps ${STEPS:-3} --warmup 1 ${BENCH_ARGS:-} \
      > "$OUT/bench_${eng}_${gib}_${chunk}.json" 2> "$OUT/bench_${eng}_${gib}_${chunk}.err"
  echo "bench $spec exit $?" | tee -a "$OUT/summary.txt"
  tail -1 "$OUT/bench_${eng}_${gib}_${chunk}.json" | tee -a "$OUT/summary.txt"
  tail -3 "$OUT/bench_${eng}_${gib}_${chunk}.err" >> "$OUT/summary.txt"
done

if [ "${SKIP_PROF:-0}" != "1" ]; then
  echo "== rocprofv3 kernel trace" | tee -a "$OUT/summary.txt"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o trace -- \
      python "$OLDPWD/bench.py" --engine ${PROF_ENGINE:-hot} --gib ${PROF_GIB:-8} --steps 3 --warmup 1 --no-cpu-baseline \
      > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err")
  echo "rocprof exit $?" | tee -a "$OUT/summary.txt"
  find "$OUT/prof" -name "*kernel_stats*" | head -3 | while read f; do echo "-- $f"; head -12 "$f"; done | tee -a "$OUT/summary.txt"
  find "$OUT/prof" -name "*kernel_trace.csv" -size +2M -delete
  # keep only the small summaries
  find "$OUT/prof" -type f ! -name "*stats*" -size +2M -delete
fi
echo "== done" | tee -a "$OUT/summary.txt"
