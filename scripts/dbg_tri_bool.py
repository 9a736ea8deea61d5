#!/usr/bin/env python3
"""The shallow-skip walks (k_tri_walk) against the prefix filters on the same inputs, repeated: records compared by CRC.
Run once with the product library and once with ACGPU_LIB=.../libacgpu_tribool.so (`make exp-tribool`: lane flags as `bool`,
loop conditions as plain __any()) -- the shape in which round 3 saw counts off by 1e-4.  usage: dbg_tri_bool.py [reps]"""
import os, sys, json, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_amd as ac
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
out = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
res = {"lib": os.environ.get("ACGPU_LIB", "libacgpu.so")}
for name, npat, seed, kind, mib in (("cnfa walk, 100k patterns", 100000, 0xAC04, ac.AhoCorasickKind.ContiguousNFA, 512),
                                    ("dfa walk, 1k patterns", 1000, 0xAC01, ac.AhoCorasickKind.DFA, 2048),
                                    ("dfa walk, 10k patterns", 10000, 0xAC07, ac.AhoCorasickKind.DFA, 512)):
    pats = ac.gen_patterns(npat, seed=seed)
    n = mib << 20
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    ac.gen_haystack(buf, offset=0, seed=0xAC02)
    for k, pos in enumerate(range(4093, n - 64, n // 997)):     # planted occurrences of every kind of depth
        p = pats[(7 * k) % len(pats)]
        buf[pos:pos + len(p)] = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
    ref = ac.AhoCorasick.builder().kind(kind).build(pats)             # default engine (a prefix filter)
    m0, _ = ref.overlapping_device(buf, out=out)
    c0 = zlib.crc32(out[: int(m0) * 24].cpu().numpy().tobytes())
    a = ac.AhoCorasick.builder().kind(kind).gpu_engine("walk").build(pats)
    bad, counts = 0, set()
    for i in range(reps):
        m, _ = a.overlapping_device(buf, out=out)
        c = zlib.crc32(out[: int(m) * 24].cpu().numpy().tobytes())
        counts.add(int(m))
        bad += int(m != m0 or c != c0)
    res[name] = {"records_filter": int(m0), "records_walk": sorted(counts), "runs": reps, "runs_that_differ": bad}
    del buf
print(json.dumps(res), flush=True)
