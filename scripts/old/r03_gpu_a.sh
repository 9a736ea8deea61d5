#!/bin/bash
# round 3, late GPU call A: full -m gpu suite, then the LDS walk (tail prefetch, lane chunk, a-z wide rows) and the
# large-set filter's bit-table gate, A/B through their environment knobs.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log
timeout 90 python scripts/bench_hot.py --steps 6 2>&1 | tail -1 | tee $O/hot_ascii.json
ACGPU_LW_LANE_CHUNK=1024 timeout 90 python scripts/bench_hot.py --steps 6 2>&1 | tail -1 | tee $O/hot_ascii_1024.json
timeout 90 python scripts/bench_hot.py --steps 6 --alpha az 2>&1 | tail -1 | tee $O/hot_az.json
timeout 150 python scripts/run_c4.py 8 auto 5 2>&1 | tail -1 | tee $O/c4_gate.json
ACGPU_PFX_GATE=0 timeout 150 python scripts/run_c4.py 8 auto 5 2>&1 | tail -1 | tee $O/c4_nogate.json
timeout 200 scripts/pmc_hot.sh 8 ascii sq3 tc3 2>&1 | tail -30
