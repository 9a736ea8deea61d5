#!/bin/bash
# round 4, GPU call A: the two parked kernels of round 3 (8-byte level 1 of the large-set filter; case-folded keys of the
# two-type filter) -- parity with each on, A/B, then the full suite.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04a; mkdir -p $O
ACGPU_PFX_KEY8=1 timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_bench_defs.py -m gpu -x -q > $O/pytest_key8.log 2>&1; echo "key8 pytest exit $?"; tail -3 $O/pytest_key8.log
timeout 200 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/key8_ab.jsonl
timeout 120 python scripts/bench_c5.py 2>&1 | tail -2 | tee $O/c5_fold.jsonl
ACGPU_PF_FOLD=0 timeout 120 python scripts/bench_c5.py 2>&1 | tail -2 | tee $O/c5_nofold.jsonl
timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log
