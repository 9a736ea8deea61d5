#!/bin/bash
# 8-byte level 1: same-box A/B of the verifier variants (t0 = one ring per round, two walks; default = merged rounds, four
# walks between rounds; t2 = three walks), then the clock profile of t0's and the default's structure (-DPFX_PROF=1 builds)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z12; mkdir -p $O; E=$PWD/aho-corasick_amd/lib/exp
for rep in 1 2; do
  ACGPU_LIB=$E/libacgpu_t0.so KEY8_VARIANTS=12 timeout 200 python scripts/key8_ab.py 1024 one 2>&1 | tail -1 | cut -c1-220 | tee -a $O/t0.jsonl
  KEY8_VARIANTS=12 timeout 200 python scripts/key8_ab.py 1024 one 2>&1 | tail -1 | cut -c1-220 | tee -a $O/t1.jsonl
  ACGPU_LIB=$E/libacgpu_t2.so KEY8_VARIANTS=12 timeout 200 python scripts/key8_ab.py 1024 one 2>&1 | tail -1 | cut -c1-220 | tee -a $O/t2.jsonl
done
ACGPU_LIB=$E/libacgpu_p0.so KEY8_VARIANTS=12,14 timeout 200 python scripts/pfx_prof.py 2>&1 | tail -4 | tee $O/p0.jsonl
ACGPU_LIB=$E/libacgpu_p1.so KEY8_VARIANTS=12,14 timeout 200 python scripts/pfx_prof.py 2>&1 | tail -4 | tee $O/p1.jsonl
