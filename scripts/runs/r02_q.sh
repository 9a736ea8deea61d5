#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02q
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_corpora.py -x -q -k "large_set or corpora" > "$OUT/pytest.log" 2>&1
echo "exit $?" | tee "$OUT/summary.txt"; tail -3 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
prof() {  # name, cmd...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/$name" -o t -- "$@" > "$OLDPWD/$OUT/$name.log" 2>&1)
  echo "== $name" | tee -a "$OUT/summary.txt"
  grep -v amdgpu "$OUT/$name.log" | grep '^{' | cut -c1-250 | tee -a "$OUT/summary.txt"
  f=$(find "$OUT/$name" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -8 "$f" | cut -c1-200 | tee -a "$OUT/summary.txt"
}
ACGPU_PFX_MIN_PATTERNS=1 prof words5000 python "$PWD/scripts/bench_inputs.py" --engines pf --only words-5000
ACGPU_PFX_MIN_PATTERNS=1 prof dict15 python "$PWD/scripts/bench_inputs.py" --engines pf --only dictionary-15
prof c4 python "$PWD/scripts/bench_c4.py" 8 100000
