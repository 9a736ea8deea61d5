#!/bin/bash
# final code of round 4 (chain tails + two hit queues, order pass with one atomic per event): the whole GPU suite, bench + kernel
# traces, natural-text PMC passes, the reference's benchmark definitions
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04z19
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04z19/pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r04z19/pytest.log
OUT=gpurun_out/r04run5 NO_PMC=1 bash scripts/r04_round.sh 2>&1 | tail -8 | cut -c1-400
timeout 300 scripts/pmc_nat.sh > gpurun_out/r04run5/pmc_nat.log 2>&1; tail -3 gpurun_out/r04run5/pmc_nat.log; cp gpurun_out/pmc_nat/pmc.json gpurun_out/r04run5/nat_pmc.json
