#!/bin/bash
# Role-split experiments on k_pfx_count: libacgpu variants with -DPFX_PRODUCERS=P -DPFX_VERIFIERS=V in lib/exp/.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
EXP=$ROOT/aho-corasick_amd/lib/exp
mkdir -p "$EXP"
make -C aho-corasick_amd/csrc -j8 > /dev/null || exit 1
for pv in "$@"; do
  P=${pv%:*}; V=${pv#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPFX_PRODUCERS=$P -DPFX_VERIFIERS=$V -I$ROOT/include -I$ROOT/aho-corasick_amd/csrc \
      -I$ROOT/aho-corasick_amd/csrc/device -c aho-corasick_amd/csrc/device/pfx_scan.hip -o "$EXP/pfx_${P}_$V.o" || exit 1
  objs=$(ls aho-corasick_amd/lib/obj/device/*.o aho-corasick_amd/lib/obj/*.o aho-corasick_amd/lib/obj/host/*.o | grep -v pfx_scan)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$EXP/libacgpu_pfx_${P}_$V.so" $objs "$EXP/pfx_${P}_$V.o" -ldl || exit 1
  rm -f "$EXP/pfx_${P}_$V.o"
  echo "built $P producers / $V verifiers"
done
