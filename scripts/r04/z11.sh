#!/bin/bash
# 8-byte level 1, merged verifier rounds, level 3 between rounds: four (default), three, two walks per lane; 12 + 4 and 14 + 2
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z11; mkdir -p $O
KEY8_VARIANTS=12,14 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/d256.jsonl
for D in 192 128; do ACGPU_LIB=$PWD/aho-corasick_amd/lib/exp/libacgpu_d$D.so KEY8_VARIANTS=12,14 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/d$D.jsonl; done
timeout 200 python scripts/bench_c4.py 8 2>&1 | tail -1 | tee $O/c4.jsonl
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_guard.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
