#!/usr/bin/env python3
"""Config 4 (100 000 patterns, reference kind contiguous NFA) over a resident haystack: default engine, whole call and
count kernel; optional pattern count / size."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [100000]
n = int(gib * (1 << 30))
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.gen_haystack(buf, offset=0, seed=0xAC02)
out = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for npat in sizes:
    pats = ac.gen_patterns(npat, seed=0xAC04)
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA).build(pats)
    p = _lib.CProfile()
    for _ in range(2):
        m, ok = a.overlapping_device(buf, out=out, profile=p)
    torch.cuda.synchronize()
    ks, t0 = [], time.perf_counter()
    for _ in range(5):
        m, ok = a.overlapping_device(buf, out=out, profile=p)
        ks.append(p.ms_scan)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    k = float(np.mean(ks))
    print(json.dumps({"patterns": npat, "gib": gib, "matches": int(m), "call_ms": round(dt * 1e3, 3), "call_GBps": round(n / dt / 1e9, 1),
                      "kernel_ms": round(k, 3), "kernel_GBps": round(n / k / 1e6, 1), "fill_ms": round(p.ms_fill, 3), "engine": int(p.engine_used),
                      "lib": os.path.basename(os.environ.get("ACGPU_LIB", "libacgpu.so"))}), flush=True)
