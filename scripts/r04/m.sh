#!/bin/bash
# round 4, GPU call M: dense large-set scans routed to the walk (split probe again), c5 with the direct selection output,
# then the whole -m gpu suite
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04m; mkdir -p $O
timeout 300 python scripts/split_probe.py 256 2>&1 | grep '"set"' | tee $O/split_probe.jsonl
timeout 120 python scripts/bench_c5.py 2>&1 | tail -2 | tee $O/c5.jsonl
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log
