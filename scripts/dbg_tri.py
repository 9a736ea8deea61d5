#!/usr/bin/env python3
"""Debug: contiguous-NFA walk engine (k_cnfa_tri) vs the oracle on small inputs. usage: dbg_tri.py npat mib"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_amd as ac
from oracle import orc
npat = int(sys.argv[1]); mib = float(sys.argv[2])
pats = orc.gen_patterns(npat, seed=0xAC04)
n = int(mib * (1 << 20))
hay = orc.gen_haystack(0, n, seed=7)
rng = np.random.default_rng(1)
for at in range(3, n - 64, 4999):
    p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8); hay[at:at + len(p)] = p
a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA).gpu_engine("walk").build(pats)
got = a.find_overlapping_iter(torch.from_numpy(hay).cuda(), as_numpy=True)
want = orc.Oracle(pats, kind=orc.KIND_CNFA).find_overlapping_iter(hay, as_numpy=True)
ok = len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in ("pattern", "start", "end"))
print("npat", npat, "mib", mib, "got", len(got), "want", len(want), "ok", ok, "guard", ac.load_library().acgpu_guard_violations(), flush=True)
