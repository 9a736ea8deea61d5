#!/usr/bin/env python3
"""Debug: test_gpu_dfa_fill case c2 step by step.  usage: dbg_c2.py <set index> <step>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_amd as ac
from oracle import orc
si, step = int(sys.argv[1]), sys.argv[2]
engine = sys.argv[3] if len(sys.argv) > 3 else "auto"
rng = np.random.default_rng(3)
sets = [orc.gen_patterns(1000, seed=0xAC01),
        [bytes(rng.integers(0x61, 0x64, size=int(rng.integers(1, 9)), dtype=np.uint8)) for _ in range(300)],
        [b"a", b"ab", b"abc", b"bc", b"c", b"abcd" * 40], []]
pats = sets[si]
hay = orc.gen_haystack(0, 20000, seed=5, lo=0x61, span=4)
d = torch.from_numpy(hay).cuda()
if step == "ov":
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).ascii_case_insensitive(True).gpu_engine(engine).build(pats)
    want = orc.Oracle(pats, kind=orc.KIND_DFA, ascii_case_insensitive=True).find_overlapping_iter(hay, as_numpy=True)
    p = ac._lib.CProfile()
    got = a.find_overlapping_iter(d, as_numpy=True, profile=p)
    ok = len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in ("pattern", "start", "end"))
    print("set", si, "overlapping", len(got), len(want), ok, "engine", int(p.engine_used), "routed", int(p.routed), flush=True)
else:
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).match_kind(1).ascii_case_insensitive(True).build(pats)
    want = orc.Oracle(pats, match_kind=1, kind=orc.KIND_DFA, ascii_case_insensitive=True).find_iter(hay, as_numpy=True)
    got = a.find_iter(d, as_numpy=True)
    print("set", si, "find_iter", len(got), len(want), np.array_equal(got["start"], want["start"]), flush=True)
