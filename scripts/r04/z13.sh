#!/bin/bash
# walk-loop trips and flush clocks of the verifiers (8-byte level 1, one ring per round, two walks per lane)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z13; mkdir -p $O; E=$PWD/aho-corasick_amd/lib/exp
ACGPU_LIB=$E/libacgpu_p0.so KEY8_VARIANTS=12 timeout 200 python scripts/pfx_prof.py 2>&1 | tail -2 | tee $O/p0.jsonl
