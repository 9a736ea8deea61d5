#!/bin/bash
# round 4, GPU call U: LDS walk with two chains per lane (64-byte units) on the odd-stride tables
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04u; mkdir -p $O
timeout 100 python scripts/bench_hot.py --steps 8 2>&1 | tail -1 | tee $O/one_chain.json
ACGPU_LW_CHAINS=2 ACGPU_LW_UNIT=64 timeout 100 python scripts/bench_hot.py --steps 8 2>&1 | tail -1 | tee $O/two_chains.json
ACGPU_LW_UNIT=64 timeout 100 python scripts/bench_hot.py --steps 8 2>&1 | tail -1 | tee $O/one_chain_u64.json
