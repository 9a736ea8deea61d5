#!/bin/bash
# round 4, GPU call D: 8-byte level 1 with large event batches -- two passes vs level 3 inline, wave roles
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04d; mkdir -p $O
KEY8_VARIANTS=0,8,12,14 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/two_pass.jsonl
ACGPU_PFX_ONE_PASS=1 KEY8_VARIANTS=0,8,12,14 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/one_pass.jsonl
