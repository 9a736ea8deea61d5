#!/usr/bin/env python3
"""Debug: repeat the contiguous-NFA walk on one input, count runs whose result differs. usage: dbg_tri_rep.py npat mib reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zlib
import aho_corasick_amd as ac
npat = int(sys.argv[1]); mib = float(sys.argv[2]); reps = int(sys.argv[3])
pats = ac.gen_patterns(npat, seed=0xAC04)
n = int(mib * (1 << 20))
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.gen_haystack(buf, offset=0, seed=0xAC02)
out = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
ref = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA).build(pats)   # default engine (filter)
m0, _ = ref.overlapping_device(buf, out=out)
c0 = zlib.crc32(out[: int(m0) * 24].cpu().numpy().tobytes())
a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA).gpu_engine("walk").build(pats)
bad = 0
for i in range(reps):
    m, _ = a.overlapping_device(buf, out=out)
    c = zlib.crc32(out[: int(m) * 24].cpu().numpy().tobytes())
    if (m, c) != (m0, c0):
        bad += 1
        print("run", i, "differs:", m, "vs", m0, flush=True)
print("npat", npat, "mib", mib, "reps", reps, "bad", bad, "matches", int(m0), "guard", ac.load_library().acgpu_guard_violations(), flush=True)
