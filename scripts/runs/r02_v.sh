#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02v
mkdir -p "$OUT"; : > "$OUT/summary.txt"
export TMPDIR=/tmp
for lib in exp/libacgpu_pfx_12_4x4.so; do
  echo "-- $lib" | tee -a "$OUT/summary.txt"
  ACGPU_LIB=$PWD/aho-corasick_amd/lib/$lib ACGPU_PFX_MIN_PATTERNS=1 timeout 300 python scripts/bench_inputs.py --engines pf --only English 2>/dev/null | grep "words-5000\|dictionary-15\|words-15000" | cut -c1-40,150-270 | tee -a "$OUT/summary.txt"
done
(cd /tmp && ACGPU_PFX_MIN_PATTERNS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/w5000" -o t -- python "$OLDPWD/scripts/bench_inputs.py" --engines pf --only words-5000 > "$OLDPWD/$OUT/w5000.log" 2>&1)
f=$(find "$OUT/w5000" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -4 "$f" | cut -c1-60,150-260 | tee -a "$OUT/summary.txt"
