#!/usr/bin/env python3
"""Match-dense variant of the headline workload (a-z alphabet, ~1.5e5 matches per GiB): how the ordered
record materialisation scales when the result set is large.  For the record in DESIGN.md; not the headline bench."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(gib * (1 << 30))
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.gen_haystack(buf, offset=0, seed=0xAC02, lo=0x61, span=26)
out = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
pats = ac.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)
a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build(pats)
prof = _lib.CProfile()
for _ in range(3):
    m, ok = a.overlapping_device(buf, out=out, profile=prof)
torch.cuda.synchronize()
t0 = time.perf_counter()
k = 5
for _ in range(k):
    m, ok = a.overlapping_device(buf, out=out, profile=prof)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / k
print(f"a-z, 1000 patterns, {gib} GiB: {m} matches, {dt*1e3:.2f} ms = {n/dt/1e9:.0f} GB/s "
      f"(count {prof.ms_scan:.2f} ms, scan {prof.ms_compact:.2f} ms, fill {prof.ms_fill:.2f} ms)")
