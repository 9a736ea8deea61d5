#!/usr/bin/env python3
"""What would splitting a dictionary by pattern length buy?  dictionary/english/sorted.txt (123 115 words, 2 004 of them
shorter than four bytes) over prose, as ONE automaton (today: the global DFA walk) and as the two automata a split engine
would run -- the short words alone (LDS walk) and the long ones alone (large-set filter): count-kernel time, call time and
records of each.  usage: split_probe.py [mib]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
import corpora
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
words = corpora.words("dictionary-sorted")
hay = corpora.haystack("sherlock.txt")
d = torch.from_numpy(np.tile(hay, -(-n // len(hay)))[:n].copy()).cuda()
out = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for name, pats in (("all", words), ("short (< 4 bytes)", [w for w in words if len(w) < 4]), ("long (>= 4 bytes)", [w for w in words if len(w) >= 4]),
                   ("long (>= 8 bytes)", [w for w in words if len(w) >= 8])):
    a = ac.AhoCorasick.builder().build(pats)
    p = _lib.CProfile()
    m, ok = a.overlapping_device(d, out=out, profile=p)
    if not ok:
        out = None; torch.cuda.empty_cache()
        out = torch.empty(int(m) * 24 + 4096, dtype=torch.uint8, device="cuda")
    ks = []
    for _ in range(2):
        m, ok = a.overlapping_device(d, out=out, profile=p)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        m, ok = a.overlapping_device(d, out=out, profile=p); ks.append(p.ms_scan)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"set": name, "patterns": len(pats), "mib": n >> 20, "records": int(m), "records_per_byte": round(int(m) / n, 3), "engine": int(p.engine_used),
                      "routed": int(p.routed), "kernel_ms": round(float(np.mean(ks)), 3), "kernel_GBps": round(n / float(np.mean(ks)) / 1e6, 1),
                      "call_ms": round(dt * 1e3, 3), "call_GBps": round(n / dt / 1e9, 1)}), flush=True)
