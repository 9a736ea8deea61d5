#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02n
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "walk or cnfa or c4 or empty or dense_matches or nnfa" > "$OUT/pytest.log" 2>&1
echo "exit $?" | tee "$OUT/summary.txt"; tail -8 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
cat > /tmp/c4walk.py <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
n = 2 << 30
buf = torch.empty(n, dtype=torch.uint8, device="cuda"); ac.gen_haystack(buf, offset=0, seed=0xAC02)
out = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for npat in (100000, 1000):
    pats = ac.gen_patterns(npat, seed=0xAC04)
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA).gpu_engine("walk").build(pats)
    p = _lib.CProfile()
    for _ in range(2): m, ok = a.overlapping_device(buf, out=out, profile=p)
    ks = []
    for _ in range(3):
        m, ok = a.overlapping_device(buf, out=out, profile=p); ks.append(p.ms_scan)
    k = float(np.mean(ks))
    print(json.dumps({"patterns": npat, "literal": os.environ.get("ACGPU_CNFA_LITERAL"), "matches": int(m), "kernel_ms": round(k, 2), "GBps": round(n / k / 1e6, 1), "engine": int(p.engine_used)}))
PY
timeout 300 python /tmp/c4walk.py 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
ACGPU_CNFA_LITERAL=1 timeout 300 python /tmp/c4walk.py 2>&1 | grep patterns | tee -a "$OUT/summary.txt"
