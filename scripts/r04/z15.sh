#!/bin/bash
# two hit queues: one verifier round per ring (default) against one round over all rings (lib/exp/libacgpu_t3.so), same box
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z15; mkdir -p $O; E=$PWD/aho-corasick_amd/lib/exp
for rep in 1 2; do
  KEY8_VARIANTS=12 timeout 200 python scripts/key8_ab.py 2>&1 | tail -2 | cut -c1-330 | tee -a $O/new.jsonl
  ACGPU_LIB=$E/libacgpu_t3.so KEY8_VARIANTS=12,14 timeout 200 python scripts/key8_ab.py 2>&1 | tail -2 | cut -c1-330 | tee -a $O/t3.jsonl
done
ACGPU_LIB=$E/libacgpu_p3.so KEY8_VARIANTS=12 timeout 200 python scripts/pfx_prof.py 2>&1 | tail -2 | tee $O/p3.jsonl
