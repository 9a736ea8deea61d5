#!/bin/bash
# clock profile of the wave roles on config 4 (lib/exp/libacgpu_prof.so: -DPFX_PROF=1)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z17; mkdir -p $O
ACGPU_LIB=$PWD/aho-corasick_amd/lib/exp/libacgpu_prof.so timeout 200 python scripts/pfx_prof_c4.py 2>&1 | tail -1 | tee $O/c4_prof.jsonl
ACGPU_PFX_GATE=0 ACGPU_LIB=$PWD/aho-corasick_amd/lib/exp/libacgpu_prof.so timeout 200 python scripts/pfx_prof_c4.py 2>&1 | tail -1 | tee -a $O/c4_prof.jsonl
