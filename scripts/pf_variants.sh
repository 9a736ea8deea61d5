#!/bin/bash
# Timing experiments on k_pf_count: builds libacgpu variants with -DPF_EXP=<mask> into aho-corasick_amd/lib/exp/
# (travels with gpurun) and, with "run", benches each one on the GPU box.  Results are not parity-valid for mask != 0.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
EXP=$ROOT/aho-corasick_amd/lib/exp
if [ "${1:-build}" = build ]; then
  shift || true
  mkdir -p "$EXP"
  make -C aho-corasick_amd/csrc -j8 > /dev/null || exit 1
  for m in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPF_EXP=$m -I$ROOT/include -I$ROOT/aho-corasick_amd/csrc \
        -I$ROOT/aho-corasick_amd/csrc/device -c aho-corasick_amd/csrc/device/pf_scan.hip -o "$EXP/pf_$m.o" || exit 1
    objs=$(ls aho-corasick_amd/lib/obj/device/*.o aho-corasick_amd/lib/obj/*.o aho-corasick_amd/lib/obj/host/*.o | grep -v pf_scan)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$EXP/libacgpu_exp$m.so" $objs "$EXP/pf_$m.o" || exit 1
    rm -f "$EXP/pf_$m.o"
    echo "built exp $m"
  done
else
  OUT=gpurun_out/exp_$(date +%H%M%S); mkdir -p "$OUT"
  for so in "$EXP"/libacgpu_exp*.so; do
    m=$(basename "$so" .so)
    ACGPU_LIB=$so timeout 300 python bench.py --engine pf --gib 8 --steps 10 --warmup 1 --no-cpu-baseline 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'matches', d['config']['matches'])" | tee -a "$OUT/variants.txt"
  done
fi
