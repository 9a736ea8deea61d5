#!/bin/bash
# One gpurun call = a list of named steps (replaces the per-call one-off scripts of earlier rounds).
# usage: scripts/gpu_steps.sh <tag> step [step ...]      output: gpurun_out/<tag>/
# steps: tests | tests:<pytest args> | smoke | bench | bench_prof | hot_ab | hot_pmc[:alpha] | defs[:filter] | defs_all |
#        c4_ab | nat_ab | fuzz[:n] | c4_pmc | nat_pmc | minlen
set -u
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
log() { echo "$@" | tee -a "$OUT/summary.txt"; }
hot() {   # hot <label> -- <bench_hot args (engine forms: --variant name=value)>
  local label=$1; shift
  while [ "$1" != "--" ]; do shift; done; shift
  timeout 300 python scripts/bench_hot.py "$@" 2>> "$OUT/hot_ab.err" | tail -1 | sed "s/^{/{\"variant\": \"$label\", /" | tee -a "$OUT/hot_ab.jsonl" | cut -c1-400
}
for step in "$@"; do
  arg=${step#*:}; [ "$arg" = "$step" ] && arg=""
  log "== $step"
  case ${step%%:*} in
    tests)
      timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 $arg > "$OUT/pytest_gpu.log" 2>&1
      log "pytest exit $?"; tail -8 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt" ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
      log "smoke exit $?"; tail -3 "$OUT/smoke.log" | tee -a "$OUT/summary.txt" ;;
    bench)
      timeout 600 python bench.py $arg > "$OUT/bench.json" 2> "$OUT/bench.err"
      log "bench exit $?"; tail -c 400 "$OUT/bench.json" | tee -a "$OUT/summary.txt"; echo ;;
    bench_prof)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof" -o b -- \
          python "$ROOT/bench.py" $arg > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/prof.err")
      log "rocprof exit $?"
      find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
      rm -rf "$OUT/prof"; head -8 "$OUT/bench_kernel_stats.csv" | cut -c1-200 ;;
    hot_ab)
      hot "ascii default" -- --steps 10
      hot "ascii LDS class map" -- --steps 10 --variant lw_cls=0
      hot "ascii computed classes" -- --steps 10 --variant lw_cls=1
      hot "a-z default" -- --alpha az --steps 5
      hot "a-z computed classes" -- --alpha az --steps 5 --variant lw_cls=1
      hot "case-insensitive" -- --casei --steps 5
      hot "ascii 1 GiB" -- --gib 1 --steps 10 ;;
    hot_pmc)
      timeout 400 scripts/pmc_hot.sh 8 ${arg:-ascii} ${PMC_PASSES:-sq1 sq3 tc3} > "$OUT/pmc_hot_${arg:-ascii}.log" 2>&1; tail -3 "$OUT/pmc_hot_${arg:-ascii}.log"
      cp gpurun_out/pmc_hot_${arg:-ascii}/pmc.json "$OUT/hot_${arg:-ascii}_pmc.json" ;;
    defs)   # defs:<name filter>[@variant=value,...]
      dv=""; case "$arg" in *@*) dv=${arg#*@}; arg=${arg%%@*} ;; esac
      BENCH_DEFS_NO_CPU=1 timeout 900 python scripts/bench_defs.py 256 auto "$arg" $dv > "$OUT/defs_${arg//[^a-z0-9]/_}_${dv//[^a-z0-9]/_}.jsonl" 2>> "$OUT/defs.err"
      log "defs exit $? ($dv)"; python scripts/defs_table.py "$OUT/defs_${arg//[^a-z0-9]/_}_${dv//[^a-z0-9]/_}.jsonl" | tee -a "$OUT/summary.txt" ;;
    defs_prof)   # per-kernel durations of a definitions subset
      (cd /tmp && BENCH_DEFS_NO_CPU=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/dprof" -o d -- \
          python "$ROOT/scripts/bench_defs.py" 256 auto "$arg" > "$ROOT/$OUT/defs_prof.jsonl" 2> "$ROOT/$OUT/defs_prof.err")
      log "rocprof exit $?"
      find "$OUT/dprof" -name "*kernel_stats.csv" -exec cp {} "$OUT/defs_kernel_stats.csv" \;
      rm -rf "$OUT/dprof"; head -14 "$OUT/defs_kernel_stats.csv" | cut -c1-180 | tee -a "$OUT/summary.txt" ;;
    defs_all)
      timeout 1500 python scripts/bench_defs.py 256 > "$OUT/bench_defs.jsonl" 2> "$OUT/bench_defs.err"
      log "defs exit $?"; python scripts/defs_table.py "$OUT/bench_defs.jsonl" | tail -130 ;;
    fuzz)
      timeout 1200 python scripts/fuzz_gpu.py ${arg:-300} > "$OUT/fuzz.txt" 2>&1; log "fuzz exit $?"; tail -3 "$OUT/fuzz.txt" | tee -a "$OUT/summary.txt" ;;
    sh)
      bash -c "$arg" > "$OUT/sh_$(date +%H%M%S).log" 2>&1; log "sh exit $?" ;;
    *) log "unknown step $step" ;;
  esac
done
log "== done"
