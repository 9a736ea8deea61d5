#!/bin/bash
# order pass: one 64-bit atomic per event in the histogram, own_pid shortcut in the emit kernels
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z7; mkdir -p $O
KEY8_VARIANTS=12 timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/nat.jsonl
timeout 900 python -m pytest tests/test_gpu_corpora.py tests/test_gpu_bench_defs.py tests/test_gpu_parity.py tests/test_gpu_enqueue.py tests/test_gpu_find_dense.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_z7 -o z7 -- python $GRAFT_REPO_ROOT/scripts/key8_ab.py 1024 one > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_z7 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/nat_kernel_stats.csv && head -14 "$f" | cut -c1-160
