#!/usr/bin/env python3
"""Times acgpu_replace_all on a device-resident haystack (for the record in DESIGN.md; not the headline bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aho_corasick_amd as ac

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gib * (1 << 30))
pats = ac.gen_patterns(1000, seed=0xAC01)
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.gen_haystack(buf, offset=0, seed=0xAC02)
for j in range(4096):   # make it worth replacing: plant occurrences all over
    p = pats[j % len(pats)]
    pos = (j + 1) * (n // 4100)
    buf[pos:pos + len(p)] = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
repl = [b"<%d>" % i for i in range(len(pats))]
for mk in (ac.MatchKind.Standard, ac.MatchKind.LeftmostFirst):
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).match_kind(mk).build(pats)
    out = a.replace_all_bytes(buf, repl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 5
    for _ in range(k):
        out = a.replace_all_bytes(buf, repl)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    print(f"{mk.name}: {gib} GiB in {dt*1e3:.2f} ms = {n/dt/1e9:.0f} GB/s of haystack (out {out.numel()} B)")
