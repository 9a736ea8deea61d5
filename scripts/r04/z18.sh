#!/bin/bash
# (the ACGPU_PFX_ROLES knob and the C4_ROLES loop of bench_c4.py existed for this measurement only: no gain, removed)
# config 4: wave roles of the gated 4-byte filter (its clock profile says the verifiers are the bottleneck: producers wait 35 %)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z18; mkdir -p $O
C4_ROLES=12,10,8,12,10,8 timeout 300 python scripts/bench_c4.py 8 2>&1 | tail -6 | tee $O/c4_roles.jsonl
