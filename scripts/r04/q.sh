#!/bin/bash
# round 4, GPU call Q: fuzzer on the final kernels (per-start table forced on half of the seeds), c5 timing
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04q; mkdir -p $O
timeout 200 python scripts/fuzz_gpu.py 150 41000 > $O/fuzz.log 2>&1; echo "fuzz exit $?"; tail -3 $O/fuzz.log
timeout 120 python scripts/bench_c5.py 2>&1 | tail -2 | tee $O/c5.jsonl
