#!/bin/bash
# (the 10 + 5 instantiation of the every-other-position form existed for this call only: 0.498 against 0.477 ms, removed)
# every-other-position level 1: 10 + 5 wave roles against 12 + 4 (the producers got a third cheaper)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04z27
KEY8_VARIANTS=12,10 timeout 60 python scripts/key8_ab.py 2>&1 | tail -2 | cut -c1-500 | tee gpurun_out/r04z27/x2_roles.jsonl
