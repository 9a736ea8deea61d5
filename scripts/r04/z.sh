#!/bin/bash
# round 4, GPU call Z: synchronous dense calls through the enqueue machinery (one synchronisation); natural text, suite
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z; mkdir -p $O
timeout 200 python scripts/bench_nat.py 10 2>&1 | grep '^{' | cut -c1-330 | tee $O/nat.jsonl
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log
