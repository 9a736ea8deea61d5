#!/bin/bash
# 8-byte level 1: ring entries of four bytes (the position alone), 256 per ring instead of 128: same-box A/B against the previous
# library (lib/exp/libacgpu_prev.so = main before this change, built in a scratch worktree), parity of the filter's callers
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z23; mkdir -p $O; E=$PWD/aho-corasick_amd/lib/exp
for rep in 1 2; do
  ACGPU_LIB=$E/libacgpu_prev.so KEY8_VARIANTS=12 timeout 200 python scripts/key8_ab.py 2>&1 | tail -2 | cut -c1-330 | tee -a $O/prev.jsonl
  KEY8_VARIANTS=12,14 timeout 200 python scripts/key8_ab.py 2>&1 | tail -2 | cut -c1-440 | tee -a $O/new.jsonl
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_guard.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
