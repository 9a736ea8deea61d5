#!/usr/bin/env python3
"""Throughput of the overlapping scan vs pattern-set size (for the record in DESIGN.md; not the headline bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(gib * (1 << 30))
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.gen_haystack(buf, offset=0, seed=0xAC02)
out = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for npat in (100, 1000, 3000, 6000, 12000, 30000, 50000, 65000, 100000):
    pats = ac.gen_patterns(npat, seed=0xAC01)
    a = ac.AhoCorasick.builder().match_kind(ac.MatchKind.Standard).build(pats)   # default kind selection
    prof = _lib.CProfile()
    for _ in range(2):
        m, ok = a.overlapping_device(buf, out=out, profile=prof)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 5
    for _ in range(k):
        m, ok = a.overlapping_device(buf, out=out, profile=prof)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    print(f"{npat:7d} patterns kind={a.kind().name:15s} engine={int(prof.engine_used)} matches={m:8d} "
          f"{dt*1e3:8.2f} ms  {n/dt/1e9:7.0f} GB/s")
