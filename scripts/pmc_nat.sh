#!/bin/bash
# PMC passes for the large-set filter on natural text (sherlock.txt / words-5000, 1 GiB); separate passes, --kernel-trace only.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc_nat
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
run_pass() {
  local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -o pmc -- \
      python "$ROOT/scripts/bench_nat.py" 1 > "$ROOT/$OUT/$name.json" 2> "$ROOT/$OUT/$name.err")
  echo "$name exit $?"
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run_pass sq2 SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH
run_pass tc1 TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
run_pass tc2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run_pass tc3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run_pass tc4 GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
find "$OUT" -name "*kernel_trace.csv" -size +1M -delete
find "$OUT" -name "*agent_info.csv" -delete
python scripts/pmc_to_json.py "$OUT" "k_pfx_count<true, 12, 4, false, true>" "$OUT/pmc.json" "per-dispatch averages of k_pfx_count<true,12,4,false,true> (8-byte level 1, level 3 inline) over natural text (sherlock.txt tiled to 1 GiB / words-5000, en-huge / words-15000); separate rocprofv3 --pmc passes (scripts/pmc_nat.sh)" | tail -45
