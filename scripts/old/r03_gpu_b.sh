#!/bin/bash
# round 3, late GPU call B: where the bit-table gate of the large-set filter loses (per-kernel times), gate + inline
# level 3, a-z on the LDS walk with 1 024-byte lane chunks, and the changed tests.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03b; mkdir -p $O
ROOT=$PWD
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "large_set_filter_with_verifier or wide_row" 2>&1 | tail -3
(cd /tmp && ACGPU_PFX_GATE=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/gate_stats -o s -- python $ROOT/scripts/run_c4.py 8 auto 5 > $ROOT/$O/c4_gate_prof.json 2>$ROOT/$O/gate_prof.err); tail -1 $O/c4_gate_prof.json
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/nogate_stats -o s -- python $ROOT/scripts/run_c4.py 8 auto 5 > $ROOT/$O/c4_nogate_prof.json 2>$ROOT/$O/nogate_prof.err); tail -1 $O/c4_nogate_prof.json
ACGPU_PFX_GATE=1 ACGPU_PFX_ONE_PASS=1 timeout 150 python scripts/run_c4.py 8 auto 5 2>&1 | tail -1 | tee $O/c4_gate_onepass.json
ACGPU_PFX_ONE_PASS=1 timeout 150 python scripts/run_c4.py 8 auto 5 2>&1 | tail -1 | tee $O/c4_nogate_onepass.json
timeout 90 python scripts/bench_hot.py --steps 6 --alpha az 2>&1 | tail -1 | tee $O/hot_az_1024.json
ACGPU_LW_LANE_CHUNK=512 timeout 90 python scripts/bench_hot.py --steps 6 --alpha az 2>&1 | tail -1 | tee $O/hot_az_512.json
for d in gate_stats nogate_stats; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -8 "$f" | cut -c1-200; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
