#!/bin/bash
# PMC passes for the dominant kernel (separate passes; --kernel-trace only, never sys/hip traces).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc_$(date +%H%M%S)
mkdir -p "$OUT"
export TMPDIR=/tmp
ENGINE=${PMC_ENGINE:-pf}
GIB=${PMC_GIB:-8}
run_pass() {  # name, counters...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OLDPWD/$OUT/$name" -o pmc -- \
      python "$OLDPWD/bench.py" --engine "$ENGINE" --gib "$GIB" --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} \
      > "$OLDPWD/$OUT/$name.json" 2> "$OLDPWD/$OUT/$name.err")
  echo "$name exit $?"
  python - "$OUT/$name" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:80]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, v in agg.items():
        if "count" in k or "fill" in k or "lw_" in k:
            print(k, dict(v))
PY
}
[ "${LIST:-0}" = "1" ] && (rocprofv3 -L > "$OUT/counters.txt" 2>&1; grep -c "" "$OUT/counters.txt")
PASSES=${PASSES:-sq1 sq2 tcc1 tcc2 tcc3}
[ "${ONLY_TCC3:-0}" = "1" ] && PASSES=tcc3
for p in $PASSES; do
  case $p in
    sq1) run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS ;;
    sq2) run_pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM ;;
    sq3) run_pass sq3 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VALU ;;
    tcc1) run_pass tcc1 FETCH_SIZE GRBM_GUI_ACTIVE ;;
    tcc2) run_pass tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum ;;
    tcc3) run_pass tcc3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum ;;
  esac
done
find "$OUT" -name "*kernel_trace.csv" -size +1M -delete
ls -R "$OUT" | head -40
