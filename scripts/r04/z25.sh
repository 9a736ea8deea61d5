#!/bin/bash
# the 8-byte level 1 probing every other position (one hash for two starts; sets of 9-byte patterns and longer): A/B against
# the every-position form ("12p" = ACGPU_PFX_KEY8_X2=0), parity of the large-set filter's callers
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z25; mkdir -p $O
KEY8_VARIANTS=12p,12,12p,12 timeout 100 python scripts/key8_ab.py 2>&1 | tail -2 | cut -c1-600 | tee $O/x2_ab.jsonl
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_guard.py -m gpu -x -q -k "long_prefix or corpora or corpus or guard or natural or words" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
