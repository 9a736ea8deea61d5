#!/bin/bash
# Round-2 evidence run: full GPU suite, smoke, default bench, rocprofv3 kernel trace of the same command, PMC passes.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02round
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== pytest -m gpu" | tee "$OUT/summary.txt"
  timeout 1700 python -m pytest tests -m gpu -x -q --timeout 900 > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -8 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
  echo "smoke exit $?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
fi
echo "== bench" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench exit $?" | tee -a "$OUT/summary.txt"; tail -c 600 "$OUT/bench.json" | tee -a "$OUT/summary.txt"
echo "== rocprofv3 kernel trace of python bench.py (and of --no-also)" | tee -a "$OUT/summary.txt"
for variant in full noalso; do
  extra=""; [ $variant = noalso ] && extra="--no-also --no-cpu-baseline"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_$variant" -o trace -- \
      python "$OLDPWD/bench.py" $extra > "$OLDPWD/$OUT/prof_${variant}_bench.json" 2> "$OLDPWD/$OUT/prof_$variant.err")
  echo "rocprof $variant exit $?" | tee -a "$OUT/summary.txt"
  python - "$OUT" $variant <<'PY'
import csv, glob, sys, collections
out, variant = sys.argv[1], sys.argv[2]
for f in glob.glob(out + f"/prof_{variant}/**/*kernel_trace.csv", recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out + f"/kernel_durations_{variant}.csv", "w") as g:
        g.write("kernel,launches,avg_ns,min_ns,max_ns,last100_avg_ns\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            last = v[-100:]
            g.write('"%s",%d,%.0f,%d,%d,%.0f\n' % (k[:100].replace('"', "'"), len(v), sum(v) / len(v), min(v), max(v), sum(last) / len(last)))
    for k, v in per.items():
        if "k_pf_count" in k and variant == "noalso":
            with open(out + "/pf_count_launches.csv", "w") as g:
                g.write("launch,duration_ns\n")
                for i, d in enumerate(v): g.write(f"{i},{d}\n")
PY
  find "$OUT/prof_$variant" -name "*kernel_trace.csv" -delete
done
head -8 "$OUT/kernel_durations_noalso.csv" | cut -c1-200 | tee -a "$OUT/summary.txt"
echo "== pmc" | tee -a "$OUT/summary.txt"
PMC_ENGINE=pf PASSES="sq1 tcc3" BENCH_ARGS="--no-also" bash scripts/gpu_pmc.sh > "$OUT/pmc_pf.log" 2>&1
PMC_ENGINE=hot PASSES="sq1 sq2 tcc3" BENCH_ARGS="--no-also" bash scripts/gpu_pmc.sh > "$OUT/pmc_hot.log" 2>&1
ls -d gpurun_out/pmc_* | tail -2 | tee -a "$OUT/summary.txt"
