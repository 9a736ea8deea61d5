#!/bin/bash
# round 4, GPU call B: wave roles of the 8-byte level 1
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04b; mkdir -p $O
timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/key8_roles.jsonl
