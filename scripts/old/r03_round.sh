#!/bin/bash
# Round-3 evidence run: changed-path GPU tests, smoke, default bench, rocprofv3 kernel trace of the same command,
# PMC passes for the LDS walk and the gated large-set filter, the input table, natural text.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03round
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
echo "== pytest (changed paths)" | tee "$OUT/summary.txt"
if [ "${FULL:-0}" = "1" ]; then timeout 500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; else
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_corpora.py tests/test_gpu_bench_defs.py tests/test_gpu_enqueue.py -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; fi
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
echo "== bench" | tee -a "$OUT/summary.txt"
timeout 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench exit $?" | tee -a "$OUT/summary.txt"; tail -c 400 "$OUT/bench.json" | tee -a "$OUT/summary.txt"
echo "== rocprofv3 --kernel-trace --stats -- python bench.py" | tee -a "$OUT/summary.txt"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof" -o b -- \
    python "$ROOT/bench.py" > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/prof.err")
echo "rocprof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
echo "== natural text, input table" | tee -a "$OUT/summary.txt"
timeout 120 python scripts/bench_nat.py 10 > "$OUT/bench_nat.jsonl" 2>&1; tail -3 "$OUT/bench_nat.jsonl"
timeout 300 python scripts/bench_inputs.py --gib 1 --steps 3 > "$OUT/bench_inputs.jsonl" 2>&1; echo "inputs exit $?"
[ "${NO_PMC:-0}" = "1" ] && exit 0
echo "== PMC" | tee -a "$OUT/summary.txt"
timeout 250 scripts/pmc_hot.sh 8 ascii sq1 sq3 tc3 > "$OUT/pmc_hot.log" 2>&1; tail -3 "$OUT/pmc_hot.log"
PMC_TAG=_gate timeout 250 scripts/pmc_c4.sh auto 2 sq1 tc3 > "$OUT/pmc_c4_gate.log" 2>&1; tail -3 "$OUT/pmc_c4_gate.log"
PASSES=tcc3 PMC_GIB=8 timeout 200 scripts/gpu_pmc.sh > "$OUT/pmc_pf.log" 2>&1; tail -2 "$OUT/pmc_pf.log"
