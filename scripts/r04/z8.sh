#!/bin/bash
# per-kernel times of the natural-text step after the order-pass change (one pair, 8-byte level 1)
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); O=gpurun_out/r04z8; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && KEY8_VARIANTS=12 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/prof" -o b -- \
    python "$ROOT/scripts/key8_ab.py" 1024 one > "$ROOT/$O/ab.jsonl" 2> "$ROOT/$O/prof.err")
find "$O/prof" -name "*kernel_stats.csv" -exec cp {} "$O/nat_kernel_stats.csv" \;
rm -rf "$O/prof"
cut -c1-150 "$O/nat_kernel_stats.csv" | head -16
