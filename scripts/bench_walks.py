#!/usr/bin/env python3
"""The two global-memory transition walks (DFA rows / contiguous-NFA failure links) at several pattern counts: count
kernel throughput over a resident random haystack.  usage: bench_walks.py [gib] [npat,npat,...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1000, 10000, 100000]
n = int(gib * (1 << 30))
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.gen_haystack(buf, offset=0, seed=0xAC02)
out = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for npat in sizes:
    pats = ac.gen_patterns(npat, seed=0xAC04)
    for kind, name in ((ac.AhoCorasickKind.DFA, "dfa"), (ac.AhoCorasickKind.ContiguousNFA, "cnfa")):
        t0 = time.perf_counter()
        a = ac.AhoCorasick.builder().kind(kind).gpu_engine("walk").build(pats)
        tb = time.perf_counter() - t0
        p = _lib.CProfile()
        for _ in range(2):
            m, ok = a.overlapping_device(buf, out=out, profile=p)
        ks = []
        for _ in range(3):
            m, ok = a.overlapping_device(buf, out=out, profile=p)
            ks.append(p.ms_scan)
        k = float(np.mean(ks))
        print(json.dumps({"patterns": npat, "walk": name, "gib": gib, "matches": int(m), "build_s": round(tb, 2), "kernel_ms": round(k, 3),
                          "kernel_GBps": round(n / k / 1e6, 1), "engine": int(p.engine_used)}), flush=True)
