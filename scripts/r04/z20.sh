#!/bin/bash
# randomized differential run of the final code against the oracle (scripts/fuzz_gpu.py, 200 s, fresh seeds)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04z20
timeout 260 python scripts/fuzz_gpu.py 200 500000 > gpurun_out/r04z20/fuzz.txt 2>&1; echo "fuzz exit $?"; tail -4 gpurun_out/r04z20/fuzz.txt
