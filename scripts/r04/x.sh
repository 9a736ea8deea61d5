#!/bin/bash
# round 4, GPU call X: the DFA shallow-skip walk on a haystack without a single candidate (what the piece scan alone costs)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04x; mkdir -p $O
timeout 120 python scripts/bench_hot.py --engine walk --steps 4 --alpha none 2>&1 | tail -1 | tee $O/dfa_walk_nocand.json
timeout 120 python scripts/bench_hot.py --engine walk --steps 4 2>&1 | tail -1 | tee $O/dfa_walk.json
timeout 120 python scripts/bench_hot.py --engine hot --steps 4 --alpha none 2>&1 | tail -1 | tee $O/hot_nocand.json
