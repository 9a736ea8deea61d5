#!/usr/bin/env python3
"""Config 5 timing breakdown: find_iter (casei LeftmostFirst) over the resident 8 GiB haystack, host vs device output."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
pats = ac.gen_patterns(1000, seed=0xAC01)
a5 = (ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).match_kind(ac.MatchKind.LeftmostFirst).ascii_case_insensitive(True).build(pats))
n = 8 << 30
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.gen_haystack(buf, offset=0, seed=0xAC02)
out = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
p = _lib.CProfile()
for mode in ("device", "host"):
    ts = []
    for i in range(8):
        torch.cuda.synchronize(); t = time.perf_counter()
        if mode == "device":
            m = a5.find_iter_device(buf, out, profile=p)[0]
        else:
            m = len(a5.find_iter(buf, as_numpy=True, profile=p))
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(json.dumps({"mode": mode, "matches": int(m), "ms": [round(x, 3) for x in ts], "ms_scan": round(p.ms_scan, 3),
                      "ms_compact": round(p.ms_compact, 3), "ms_fill": round(p.ms_fill, 3), "ms_total": round(p.ms_total, 3),
                      "engine": int(p.engine_used)}))
