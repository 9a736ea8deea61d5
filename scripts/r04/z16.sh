#!/bin/bash
# two hit queues, 12-byte LDS events: wave roles 12 + 4 / 10 + 5 / 14 + 2, rounds from 96 survivors (lib/exp/libacgpu_m96.so); parity
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z16; mkdir -p $O; E=$PWD/aho-corasick_amd/lib/exp
for rep in 1 2; do
  KEY8_VARIANTS=12,10,14 timeout 200 python scripts/key8_ab.py 2>&1 | tail -2 | cut -c1-440 | tee -a $O/new.jsonl
  ACGPU_LIB=$E/libacgpu_m96.so KEY8_VARIANTS=12,10 timeout 200 python scripts/key8_ab.py 2>&1 | tail -2 | cut -c1-440 | tee -a $O/m96.jsonl
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_guard.py tests/test_gpu_bench_defs.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
