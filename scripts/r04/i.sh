#!/bin/bash
# round 4, GPU call I: kernel breakdown of the per-start table path
set -u
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $R/scripts/bench_defs.py 256 auto same/onebyte-match,teddy1-16pat-common > $O/run.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-200
find $O/prof -type f ! -name "*stats*" -delete
grep '"bench"' $O/run.log | cut -c1-300
