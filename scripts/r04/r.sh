#!/bin/bash
# round 4, GPU call R: whole suite on the final kernels, then the evidence run
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04r; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
bash scripts/r04_round.sh > $O/round.log 2>&1; tail -12 $O/round.log | cut -c1-300
