#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02g
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== A/B routing on the headline (enqueue form)" | tee "$OUT/summary.txt"
for i in 1 2 3; do
  for r in 0 1; do
    if [ $r = 1 ]; then export ACGPU_NO_ROUTING=1; else unset ACGPU_NO_ROUTING; fi
    timeout 300 python bench.py --no-also --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_routing=$r', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])" | tee -a "$OUT/summary.txt"
  done
done
unset ACGPU_NO_ROUTING
echo "== rocprofv3 kernel trace of python bench.py" | tee -a "$OUT/summary.txt"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o trace -- \
    python "$OLDPWD/bench.py" > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err")
echo "rocprof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -3 | while read f; do echo "-- $f"; head -30 "$f"; done | tee -a "$OUT/summary.txt"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    with open(out + "/kernel_durations.csv", "w") as g:
        g.write("kernel,launches,avg_ns,min_ns,max_ns,last100_avg_ns\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            last = v[-100:]
            g.write('"%s",%d,%.0f,%d,%d,%.0f\n' % (k[:120], len(v), sum(v) / len(v), min(v), max(v), sum(last) / len(last)))
PY
head -20 "$OUT/kernel_durations.csv" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_trace.csv" -size +4M -delete
