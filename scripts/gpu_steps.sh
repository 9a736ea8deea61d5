#!/bin/bash
# One gpurun call = a list of named steps (replaces the per-call one-off scripts of earlier rounds).
# usage: scripts/gpu_steps.sh <tag> step [step ...]      output: gpurun_out/<tag>/
# steps: tests | tests:<pytest args> | smoke | bench | bench_prof | hot_ab | hot_pmc[:alpha] | defs[:filter] | defs_all |
#        c4_ab | nat_ab | fuzz[:n] | c4_pmc | nat_pmc | minlen | trace:<defs>[@variant=v] | latency[:max MiB] | evidence
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
    c4_ab)   # config 4: default engine and the named walk, 8 GiB (crc of the records printed: variants must agree)
      for v in ${arg:-default}; do
        lib=""; [ "$v" != default ] && lib="ACGPU_LIB=$ROOT/aho-corasick_amd/lib/exp/libacgpu_pfx_$v.so"
        env $lib timeout 300 python scripts/run_c4.py 8 auto 5 2>> "$OUT/c4.err" | tail -1 | sed "s/^{/{\"variant\": \"$v\", /" | tee -a "$OUT/c4_ab.jsonl"
      done
      timeout 300 python scripts/run_c4.py 8 walk 2 2>> "$OUT/c4.err" | tail -1 | sed "s/^{/{\"variant\": \"walk\", /" | tee -a "$OUT/c4_ab.jsonl" ;;
    minlen)
      timeout 900 python scripts/minlen_sweep.py ${arg:-1} > "$OUT/minlen_sweep.jsonl" 2> "$OUT/minlen.err"; log "minlen exit $?"; cut -c1-330 "$OUT/minlen_sweep.jsonl" | tee -a "$OUT/summary.txt" ;;
    evidence)   # the round's evidence run on the FINAL code: counter traffic per bench line first (separate --pmc passes), then the
                # bench line that cites those files, then rocprofv3 --kernel-trace --stats of the same command
      T=scripts/pmc_traffic.sh
      $T "$OUT/pf_pmc.json" "k_pf_count<false, false>" 8 "headline workload (1000 patterns, 8 GiB), k_pf_count<false,false>" -- python $ROOT/bench.py --no-also --no-cpu-baseline --steps 2 --warmup 1
      $T "$OUT/dfa_tri_pmc.json" "k_tri_walk<" 8 "headline workload, DFA walk from global tables behind the shallow skip" -- python $ROOT/scripts/bench_hot.py --engine walk --steps 2
      $T "$OUT/c4_pfx_pmc.json" "k_pfx_count<false" 8 "config 4 (100000 patterns, 8 GiB), default engine" -- python $ROOT/scripts/run_c4.py 8 auto 2
      $T "$OUT/c4_cnfa_tri_pmc.json" "k_tri_walk<" 8 "config 4, contiguous-NFA failure-link walk behind the shallow skip" -- python $ROOT/scripts/run_c4.py 8 walk 1
      $T "$OUT/c5_pf_pmc.json" "k_pf_count<false, true>" 8 "config 5 (casei LeftmostFirst find_iter, 8 GiB): the occurrence scan with case-folded keys" -- python $ROOT/scripts/bench_c5.py
      $T "$OUT/nat_sherlock_pmc.json" "k_pfx_count<true" 1 "sherlock.txt tiled to 1 GiB / words-5000, long-key level 1 at every other position" -- python $ROOT/scripts/bench_nat.py 4 sherlock
      $T "$OUT/nat_enhuge_pmc.json" "k_pfx_count<true" 1 "en-huge.txt tiled to 1 GiB / words-15000" -- python $ROOT/scripts/bench_nat.py 4 en-huge
      BENCH_DEFS_NO_CPU=1 $T "$OUT/sorted_txt_walk_pmc.json" "k_tri_walk<" 0.25 "dictionary/english/sorted.txt (123 115 words) over sherlock.txt tiled to 256 MiB: the count walk" -- python $ROOT/scripts/bench_defs.py 256 auto sorted.txt
      for d in teddy3-1pat-common teddy1-16pat-uncommon teddy1-1pat-common teddy1-1pat-uncommon; do
        BENCH_DEFS_NO_CPU=1 $T "$OUT/lw_ev_${d//-/_}_pmc.json" "k_lw_count_ev<" 0.25 "reference definition $d, 256 MiB: the LDS count walk with match events" -- python $ROOT/scripts/bench_defs.py 256 auto $d
      done
      timeout 400 scripts/pmc_hot.sh 8 ascii sq1 sq2 sq3 tc3 > "$OUT/pmc_hot.log" 2>&1; cp gpurun_out/pmc_hot_ascii/pmc.json "$OUT/hot_pmc.json"; tail -2 "$OUT/pmc_hot.log"
      # what follows the scan in a config-5 step: every launch with start offset, duration and the gap in front of it
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/tr_c5" -o t -- python "$ROOT/scripts/bench_c5.py" > "$ROOT/$OUT/c5_under_rocprof.txt" 2>&1)
      python scripts/step_timeline.py "$OUT/tr_c5" "k_pf_count<" > "$OUT/c5_step_timeline.txt" 2>&1; rm -rf "$OUT/tr_c5"
      # ... and in a natural-text step (the fused order chain: four launches behind the scan)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/tr_nat" -o t -- python "$ROOT/scripts/bench_nat.py" 4 sherlock > "$ROOT/$OUT/nat_under_rocprof.txt" 2>&1)
      python scripts/step_timeline.py "$OUT/tr_nat" "k_pfx_count<" > "$OUT/nat_step_timeline.txt" 2>&1; rm -rf "$OUT/tr_nat"
      # (the bench lines below cite these files: the same code, the same box)
      for f in pf dfa_tri c4_pfx c4_cnfa_tri c5_pf nat_sherlock nat_enhuge sorted_txt_walk hot lw_ev_teddy3_1pat_common lw_ev_teddy1_16pat_uncommon lw_ev_teddy1_1pat_common lw_ev_teddy1_1pat_uncommon; do cp "$OUT/${f}_pmc.json" "profiles/${ROUND:-r06}_${f}_pmc.json"; done
      timeout 700 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; log "bench exit $?"; tail -c 300 "$OUT/bench.json"; echo
      for flags in "" "--no-also --no-cpu-baseline"; do
        tag=bench; [ -n "$flags" ] && tag=bench_noalso
        (cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof_$tag" -o b -- \
            python "$ROOT/bench.py" $flags > "$ROOT/$OUT/${tag}_under_rocprof.json" 2> "$ROOT/$OUT/prof_$tag.err")
        log "rocprof $tag exit $?"
        find "$OUT/prof_$tag" -name "*kernel_stats.csv" -exec cp {} "$OUT/${tag}_kernel_stats.csv" \;
        if [ "$tag" = bench_noalso ]; then
          python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/prof_bench_noalso/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_pf_count<" in r["Kernel_Name"]]
    with open(out + "/pf_count_launches.csv", "w") as w:
        w.write("launch,duration_ns\n")
        for i, r in enumerate(rows):
            w.write(f"{i},{int(r['End_Timestamp']) - int(r['Start_Timestamp'])}\n")
    print("k_pf_count launches:", len(rows))
PY
        fi
        rm -rf "$OUT/prof_$tag"
      done ;;
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
    trace)   # trace:<name,name*,...>[@variant=value,...]  per-call launch timelines of benchmark definitions
      dv=""; case "$arg" in *@*) dv=${arg#*@}; arg=${arg%%@*} ;; esac
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/tr_defs" -o t -- \
          python "$ROOT/scripts/trace_defs.py" ${TRACE_MIB:-256} "$arg" $dv > "$ROOT/$OUT/trace_calls.jsonl" 2> "$ROOT/$OUT/trace.err")
      log "trace exit $?"
      python scripts/call_timeline.py "$OUT/tr_defs" "$OUT/trace_calls.jsonl" > "$OUT/call_timelines.txt" 2>&1
      find "$OUT/tr_defs" -name "*kernel_trace.csv" -exec sh -c 'gzip -c "$1" > "$2/trace_kernel_trace.csv.gz"' _ {} "$OUT" \;
      rm -rf "$OUT/tr_defs"; grep "^===" "$OUT/call_timelines.txt" | cut -c1-200 | tee -a "$OUT/summary.txt" ;;
    latency)
      timeout 1500 python scripts/bench_latency.py ${arg:-1024} > "$OUT/latency_vs_size.jsonl" 2> "$OUT/latency.err"; log "latency exit $?"
      grep faster_than "$OUT/latency_vs_size.jsonl" | cut -c1-300 | tee -a "$OUT/summary.txt" ;;
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
