#!/bin/bash
# round 4, GPU call Y: a longer fuzzer run on the final code
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04y; mkdir -p $O
timeout 420 python scripts/fuzz_gpu.py 330 52000 > $O/fuzz.log 2>&1; echo "fuzz exit $?"; tail -4 $O/fuzz.log | cut -c1-400
