#!/usr/bin/env python3
"""Config 4 (100 000 patterns, ContiguousNFA) with a chosen device engine over a resident haystack.
usage: run_c4.py [gib] [engine auto|walk] [steps] [npat]  -- one JSON line (whole call, count kernel, fill)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
engine = sys.argv[2] if len(sys.argv) > 2 else "walk"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
npat = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
n = int(gib * (1 << 30))
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.gen_haystack(buf, offset=0, seed=0xAC02)
out = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
pats = ac.gen_patterns(npat, seed=0xAC04)
a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA).gpu_engine(engine).build(pats)
p = _lib.CProfile()
m, ok = a.overlapping_device(buf, out=out, profile=p)
torch.cuda.synchronize()
ks, fs, t0 = [], [], time.perf_counter()
for _ in range(steps):
    m, ok = a.overlapping_device(buf, out=out, profile=p)
    ks.append(p.ms_scan); fs.append(p.ms_fill)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / max(steps, 1)
k = float(np.mean(ks)) if ks else 0.0
rec = out[: int(m) * 24].cpu().numpy()
import zlib
print(json.dumps({"patterns": npat, "gib": gib, "engine_req": engine, "engine": int(p.engine_used), "matches": int(m),
                  "crc": zlib.crc32(rec.tobytes()), "call_ms": round(dt * 1e3, 3), "call_GBps": round(n / dt / 1e9, 1),
                  "kernel_ms": round(k, 3), "kernel_GBps": round(n / max(k, 1e-9) / 1e6, 1),
                  "fill_ms": round(float(np.mean(fs)) if fs else 0, 3)}), flush=True)
