#!/bin/bash
# round 4, GPU call V: wave roles of the gated large-set filter on config 4 (the gate took most of the verifiers' work away)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04v; mkdir -p $O
timeout 200 python scripts/run_c4.py 8 auto 5 2>&1 | tail -1 | tee $O/c4_12_4.json
for v in 14_2 15_1 10_5; do
  ACGPU_LIB=$PWD/aho-corasick_amd/lib/exp/libacgpu_pfx_$v.so timeout 200 python scripts/run_c4.py 8 auto 5 2>&1 | tail -1 | tee $O/c4_$v.json
done
