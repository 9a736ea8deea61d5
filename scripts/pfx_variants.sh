#!/bin/bash
# Role-split experiments on k_pfx_count: libacgpu variants with -DPFX_PRODUCERS=P -DPFX_VERIFIERS=V in lib/exp/.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
EXP=$ROOT/aho-corasick_amd/lib/exp
mkdir -p "$EXP"
make -C aho-corasick_amd/csrc -j8 > /dev/null || exit 1
# each argument: P:V or P:V:EXP (PFX_EXP timing experiments: results are WRONG by design)
for pv in "$@"; do
  P=${pv%%:*}; rest=${pv#*:}; V=${rest%%:*}; X=0; [ "$rest" != "$V" ] && X=${rest#*:}
  [ "$X" != 0 ] && V=${V}x$X
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPFX_PRODUCERS=$P -DPFX_VERIFIERS=${V%%x*} -DPFX_EXP=$X -I$ROOT/include -I$ROOT/aho-corasick_amd/csrc \
      -I$ROOT/aho-corasick_amd/csrc/device -c aho-corasick_amd/csrc/device/pfx_scan.hip -o "$EXP/pfx_${P}_$V.o" || exit 1
  objs=$(ls aho-corasick_amd/lib/obj/device/*.o aho-corasick_amd/lib/obj/*.o aho-corasick_amd/lib/obj/host/*.o | grep -v pfx_scan)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$EXP/libacgpu_pfx_${P}_$V.so" $objs "$EXP/pfx_${P}_$V.o" -ldl || exit 1
  rm -f "$EXP/pfx_${P}_$V.o"
  echo "built $P producers / $V verifiers"
done
