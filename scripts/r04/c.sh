#!/bin/bash
# round 4, GPU call C: where the natural-text time goes (k_pfx_count vs the second pass), 4-byte vs 8-byte level 1
set -u
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in 0 12; do
  KEY8_VARIANTS=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o t -- python $R/scripts/key8_ab.py 1024 one > $O/run_$v.log 2>&1
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); echo "== variant $v"; head -8 "$f" | cut -c1-220
  find $O/prof_$v -type f ! -name "*stats*" -delete
done
