#!/bin/bash
# final code of round 4: the whole GPU suite, then bench + kernel traces (no PMC passes: the counted kernels did not change)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04z9
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04z9/pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r04z9/pytest.log
OUT=gpurun_out/r04run4 NO_PMC=1 NO_DEFS=1 bash scripts/r04_round.sh 2>&1 | tail -12
