#!/bin/bash
# round 4, GPU call N: the `bool` lane-flag flavour of the shallow-skip walks beside the product library
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04n; mkdir -p $O
timeout 300 python scripts/dbg_tri_bool.py 6 2>&1 | tail -1 | tee $O/product.json
ACGPU_LIB=$PWD/aho-corasick_amd/lib/libacgpu_tribool.so timeout 300 python scripts/dbg_tri_bool.py 6 2>&1 | tail -1 | tee $O/tribool.json
