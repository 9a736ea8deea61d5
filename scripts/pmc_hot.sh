#!/bin/bash
# PMC passes for a transition-walk engine on the headline workload: the LDS walk (k_lw_count, engine "hot") by default,
# PMC_ENGINE=walk PMC_KERNEL=k_tri_walk for the global DFA walk; separate passes, --kernel-trace only.
# usage: pmc_hot.sh <gib> <alpha: ascii|az> [passes...]
set -u
cd "$(dirname "$0")/.."
GIB=${1:-8}; ALPHA=${2:-ascii}; shift 2 || true
PASSES=${*:-sq1 sq3 tc3}
ENGINE=${PMC_ENGINE:-hot}; KERN=${PMC_KERNEL:-k_lw_count}
OUT=gpurun_out/pmc_${ENGINE}_${ALPHA}${PMC_TAG:-}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
run_pass() {
  local name=$1; shift
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -o pmc -- \
      python "$ROOT/scripts/bench_hot.py" --engine "$ENGINE" --gib "$GIB" --alpha "$ALPHA" --steps 2 > "$ROOT/$OUT/$name.json" 2> "$ROOT/$OUT/$name.err")
  echo "$name exit $?"
}
for p in $PASSES; do
  case $p in
    sq1) run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS ;;
    sq2) run_pass sq2 SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH ;;
    sq3) run_pass sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU ;;
    tc1) run_pass tc1 TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum ;;
    tc4) run_pass tc4 GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum ;;
    tc3) run_pass tc3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum ;;
  esac
done
find "$OUT" -name "*kernel_trace.csv" -size +1M -delete
find "$OUT" -name "*agent_info.csv" -delete
python scripts/pmc_to_json.py "$OUT" "$KERN" "$OUT/pmc.json" "per-dispatch averages of $KERN, 1000 patterns ($ALPHA), $GIB GiB; separate rocprofv3 --pmc passes (scripts/pmc_hot.sh)" "$GIB" | tail -40
