#!/usr/bin/env python3
"""Debug: where do the walk engine's records differ from the oracle's?  usage: dbg_tri_diff.py npat mib"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_amd as ac
from oracle import orc
npat = int(sys.argv[1]); mib = float(sys.argv[2])
pats = orc.gen_patterns(npat, seed=0xAC04)
n = int(mib * (1 << 20))
hay = orc.gen_haystack(0, n, seed=0xAC02)
a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA).gpu_engine("walk").build(pats)
got = a.find_overlapping_iter(torch.from_numpy(hay).cuda(), as_numpy=True)
want = orc.Oracle(pats, kind=orc.KIND_CNFA).find_overlapping_iter(hay, as_numpy=True)
print("got", len(got), "want", len(want))
i = j = shown = 0
while i < len(got) and j < len(want) and shown < 12:
    g, w = got[i], want[j]
    if (g["pattern"], g["start"], g["end"]) == (w["pattern"], w["start"], w["end"]):
        i += 1; j += 1; continue
    # an extra (garbage) record in got, or a missing one
    if g["end"] <= w["end"] or g["end"] > n:
        e = int(want[j - 1]["end"]) if j else 0
        print("extra got[%d] = %s after end %d (chunk %d, off %d); next want end %d chunk %d" % (i, g, e, e // 2048, e % 2048, int(w["end"]), int(w["end"]) // 2048))
        i += 1
    else:
        print("missing want[%d] = %s chunk %d off %d" % (j, w, int(w["end"]) // 2048, int(w["end"]) % 2048)); j += 1
    shown += 1
