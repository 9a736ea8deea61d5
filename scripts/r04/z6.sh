#!/bin/bash
# (record of a GPU call made on branch exp/pfx-self: the "self" variant of key8_ab.py exists there)
# the 8-byte level 1 without wave roles (k_pfx_self): A/B against 12 + 4, then parity with it switched on
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z6; mkdir -p $O
KEY8_VARIANTS=12,self timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/self_ab.jsonl
ACGPU_PFX_KEY8_SELF=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_guard.py -m gpu -x -q -k "long_prefix or corpora or corpus or guard or words or natural" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
