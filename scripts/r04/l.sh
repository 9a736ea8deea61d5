#!/bin/bash
# round 4, GPU call L: contiguous-NFA walk with candidates fetched on speculation (config 4's named kernel); parity; the
# dictionary split probe
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04l; mkdir -p $O
timeout 300 python scripts/run_c4.py 8 walk 3 2>&1 | tail -1 | tee $O/c4_walk.json
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_guard.py tests/test_gpu_corpora.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
timeout 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_full.log 2>&1; echo "fullsize exit $?"; tail -3 $O/pytest_full.log
timeout 300 python scripts/split_probe.py 256 2>&1 | grep '"set"' | tee $O/split_probe.jsonl
