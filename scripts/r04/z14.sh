#!/bin/bash
# 8-byte level 1 with chain tails AND two hit queues (tail compares / walks in batches of their own): same-box A/B against the
# previous verifier (lib/exp/libacgpu_t0.so), 12 + 4 and 14 + 2 roles, clock profile, parity
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z14; mkdir -p $O; E=$PWD/aho-corasick_amd/lib/exp
for rep in 1 2; do
  ACGPU_LIB=$E/libacgpu_t0.so KEY8_VARIANTS=12 timeout 200 python scripts/key8_ab.py 2>&1 | tail -2 | cut -c1-330 | tee -a $O/t0.jsonl
  KEY8_VARIANTS=12,14 timeout 200 python scripts/key8_ab.py 2>&1 | tail -2 | cut -c1-330 | tee -a $O/new.jsonl
done
ACGPU_LIB=$E/libacgpu_p2.so KEY8_VARIANTS=12,14 timeout 200 python scripts/pfx_prof.py 2>&1 | tail -4 | tee $O/p2.jsonl
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_guard.py tests/test_gpu_bench_defs.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
