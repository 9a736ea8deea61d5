#!/bin/bash
# Round-4 evidence run: default bench, rocprofv3 kernel trace of the same command and of --no-also, PMC passes for the
# headline filter, the LDS walk (ascii, a-z) and the large-set filter on natural text, natural-text A/B, the reference's
# benchmark definitions (completed calls + CPU baselines).  Everything lands in gpurun_out/r04round; the summaries that
# are evidence are copied to profiles/r04_* by hand afterwards.
set -u
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/r04round}
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
echo "== bench" | tee "$OUT/summary.txt"
timeout 500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench exit $?" | tee -a "$OUT/summary.txt"; tail -c 300 "$OUT/bench.json" | tee -a "$OUT/summary.txt"; echo
echo "== rocprofv3 --kernel-trace --stats -- python bench.py" | tee -a "$OUT/summary.txt"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof" -o b -- \
    python "$ROOT/bench.py" > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/prof.err")
echo "rocprof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
rm -rf "$OUT/prof"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof2" -o b -- \
    python "$ROOT/bench.py" --no-also --no-cpu-baseline > "$ROOT/$OUT/bench_noalso_under_rocprof.json" 2> "$ROOT/$OUT/prof2.err")
echo "rocprof --no-also exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof2" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_noalso_kernel_stats.csv" \;
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/prof2/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_pf_count<" in r["Kernel_Name"]]
    with open(out + "/pf_count_launches.csv", "w") as w:
        w.write("launch,duration_ns\n")
        for i, r in enumerate(rows):
            w.write(f"{i},{int(r['End_Timestamp']) - int(r['Start_Timestamp'])}\n")
    print("k_pf_count launches:", len(rows))
PY
rm -rf "$OUT/prof2"
[ "${NO_PMC:-0}" = "1" ] || {
echo "== PMC" | tee -a "$OUT/summary.txt"
timeout 250 scripts/pmc_hot.sh 8 ascii sq1 sq3 tc3 > "$OUT/pmc_hot.log" 2>&1; tail -2 "$OUT/pmc_hot.log"
python scripts/pmc_to_json.py gpurun_out/pmc_hot_ascii "k_lw_count" "$OUT/hot_pmc.json" "per-dispatch averages of k_lw_count, 1000 patterns (ascii), 8 GiB, odd row stride; separate rocprofv3 --pmc passes (scripts/pmc_hot.sh)" > /dev/null
timeout 200 scripts/pmc_hot.sh 8 az sq1 sq3 > "$OUT/pmc_hot_az.log" 2>&1; tail -2 "$OUT/pmc_hot_az.log"
python scripts/pmc_to_json.py gpurun_out/pmc_hot_az "k_lw_count" "$OUT/hot_az_pmc.json" "per-dispatch averages of k_lw_count, 1000 a-z patterns, 8 GiB a-z haystack, odd row stride (wide handles); separate rocprofv3 --pmc passes" > /dev/null
BENCH_ARGS=--no-also PASSES=tcc3 PMC_GIB=8 timeout 200 scripts/gpu_pmc.sh > "$OUT/pmc_pf.log" 2>&1; tail -2 "$OUT/pmc_pf.log"
d=$(ls -d gpurun_out/pmc_[0-9]* | tail -1); python scripts/pmc_to_json.py "$d" "k_pf_count<" "$OUT/pf_pmc.json" "per-dispatch averages of k_pf_count<false,false>, headline workload 8 GiB; rocprofv3 --pmc TCC_EA0_RDREQ* pass (scripts/gpu_pmc.sh)" > /dev/null
timeout 300 scripts/pmc_nat.sh > "$OUT/pmc_nat.log" 2>&1; tail -3 "$OUT/pmc_nat.log"; cp gpurun_out/pmc_nat/pmc.json "$OUT/nat_pmc.json"
}
echo "== natural text A/B, definitions" | tee -a "$OUT/summary.txt"
KEY8_VARIANTS=0,12 timeout 200 python scripts/key8_ab.py > "$OUT/nat_ab.jsonl" 2>&1; tail -2 "$OUT/nat_ab.jsonl" | cut -c1-300
[ "${NO_DEFS:-0}" = "1" ] || { timeout 900 python scripts/bench_defs.py 256 > "$OUT/bench_defs.jsonl" 2> "$OUT/bench_defs.err"; echo "defs exit $?"; grep -c '"bench"' "$OUT/bench_defs.jsonl"; }
echo "== done" | tee -a "$OUT/summary.txt"
