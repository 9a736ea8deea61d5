#!/bin/bash
# round 4, GPU call O: wave-level compaction of the ring pushes (config 4, natural text) A/B; sticky routing; parity
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04o; mkdir -p $O
timeout 200 python scripts/run_c4.py 8 auto 5 2>&1 | tail -1 | tee $O/c4_compact.json
ACGPU_PFX_PUSH_LOOP=1 timeout 200 python scripts/run_c4.py 8 auto 5 2>&1 | tail -1 | tee $O/c4_loop.json
KEY8_VARIANTS=12 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/nat_compact.jsonl
ACGPU_PFX_PUSH_LOOP=1 KEY8_VARIANTS=12 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/nat_loop.jsonl
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py tests/test_gpu_bench_defs.py tests/test_gpu_guard.py tests/test_gpu_tri_bool_flavour.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
