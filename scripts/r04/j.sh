#!/bin/bash
# round 4, GPU call J: per-start table with cheaper doubling rounds -- parity, kernel breakdown, the dense definitions
set -u
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_find_dense.py tests/test_gpu_find.py -m gpu -x -q > $O/pytest_dense.log 2>&1; echo "dense pytest exit $?"; tail -3 $O/pytest_dense.log
export TMPDIR=/tmp
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $R/scripts/bench_defs.py 256 auto teddy1-16pat-common > $O/run.log 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep k_ss "$f" | cut -c1-40,150-260
find $O/prof -type f ! -name "*stats*" -delete
timeout 300 python scripts/bench_defs.py 256 auto onebyte-match,teddy1-1pat-common,teddy1-16pat-common,big16earlyshort,teddy3-64pat-common,sorted.txt 2>&1 | grep '"bench"' | cut -c1-400 | tee $O/defs.jsonl
