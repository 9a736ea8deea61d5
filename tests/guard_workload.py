"""Workload of tests/test_gpu_guard.py: run with ACGPU_LIB = the bounds-checked flavour of the library.  Every engine,
misaligned haystack pointers, spans of every alignment, shards, find_iter / find / replace_all / stream search, each
checked against the oracle; prints `violations <n>` (acgpu_guard_violations) at the end."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
from gpu_util import assert_same, build_pair, plant
from oracle import orc

L = _lib.load_library()
assert L.acgpu_guard_violations() == 0, "not a guard build (or violations before any search)"
rng = np.random.default_rng(11)
calls = 0


def spans(n):
    out = [(0, n), (1, n - 1), (0, 17), (n - 33, n), (63, 65), (64, 64 + 4096), (5, 5)]
    out += [tuple(sorted(int(x) for x in rng.integers(0, n + 1, size=2))) for _ in range(6)]
    return out


def check(a, o, hay, tag, big=None):
    """the haystack ends exactly at the end of its device allocation when mis == 0 (nothing behind it to read)"""
    global calls
    n = len(hay)
    for mis in (0, 1, 7, 16, 33):
        buf = torch.zeros(n + mis, dtype=torch.uint8, device="cuda")
        buf[mis:] = torch.from_numpy(hay).cuda()
        d = buf[mis:]
        for s0, s1 in spans(n):
            want = o.find_overlapping_iter(hay, span=(s0, s1), as_numpy=True)
            assert_same(a.find_overlapping_iter(ac.Input(d).range(s0, s1), as_numpy=True), want, f"{tag} mis={mis} span=({s0},{s1})")
            calls += 1
        mid = n // 2 + 3
        parts = [a.find_overlapping_shard(ac.Input(d), 0, mid), a.find_overlapping_shard(ac.Input(d), mid, n)]
        assert_same(np.concatenate(parts), o.find_overlapping_iter(hay, as_numpy=True), f"{tag} mis={mis} shards")
        calls += 2


n = (1 << 20) + 37
pats = orc.gen_patterns(1000, seed=0xAC01)
hay = orc.gen_haystack(0, n, seed=0xAC02)
plant(hay, pats[:40], [0, 50, 4090, 65530, n - 16, n - 5] + [8191 * k for k in range(1, 100)])
for engine in ("pf", "hot", "walk"):
    a, o = build_pair(pats, "standard", {"kind": "dfa"}, engine=engine)
    check(a, o, hay, f"dfa {engine}")
a, o = build_pair(pats, "standard", {"kind": "cnfa"}, engine="walk")
check(a, o, hay, "cnfa walk")
# large-set filter: 4-byte level 2, and the long-prefix level 2 (shortest pattern 6 bytes), on sparse and hit-dense input
LARGE = {"pfx_min_patterns": 1}   # engine variant: the large-set filter for every set it can serve
p30 = orc.gen_patterns(30000, seed=0xAC05)
h30 = orc.gen_haystack(0, n, seed=0xAC03)
plant(h30, p30[::200], [4099 * k for k in range(1, 200)] + [0, n - 8])
a, o = build_pair(p30, "standard", {"kind": "dfa"}, engine="pf", variants=LARGE)
check(a, o, h30, "large set")
long_p = [bytes((p * 3)[: 6 + (i % 9)]) for i, p in enumerate(orc.gen_patterns(3000, seed=0xAC07, lo=0x61, span=26))]
hl = orc.gen_haystack(0, n, seed=0xAC08, lo=0x61, span=26)
plant(hl, long_p[::20], [4099 * k for k in range(1, 200)] + [0, n - 9])
a, o = build_pair(long_p, "standard", {"kind": "dfa"}, engine="pf", variants=LARGE)
check(a, o, hl, "large set, long prefix")
pieces = [long_p[int(i)][: int(k)] for i, k in zip(rng.integers(0, len(long_p), size=200000), rng.integers(4, 15, size=200000))]
hd = np.frombuffer(b"".join(pieces), dtype=np.uint8)[: 1 << 20].copy()
check(a, o, hd, "large set, long prefix, hit-dense")
# adversarial for the two-type filter (routed to the LDS walk), dense results (classic pipeline + fills)
pre = np.frombuffer(b"".join(p[:4] for p in pats) * 300, dtype=np.uint8)[: 1 << 20].copy()
a, o = build_pair(pats, "standard", {"kind": "dfa"})
check(a, o, pre, "routed")
az = [p[:3] for p in orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)]
haz = orc.gen_haystack(0, 1 << 18, seed=0xAC02, lo=0x61, span=26)
for engine in ("pf", "hot", "walk"):
    a, o = build_pair(az, "standard", {"kind": "dfa"}, engine=engine)
    check(a, o, haz, f"dense {engine}")
# non-overlapping searches, replace_all, stream search
for mk in ("leftmost_first", "leftmost_longest", "standard"):
    a, o = build_pair(pats, mk, {"kind": "dfa"})
    for mis in (0, 5):
        buf = torch.zeros(n + mis, dtype=torch.uint8, device="cuda")
        buf[mis:] = torch.from_numpy(hay).cuda()
        for s0, s1 in spans(n)[:5]:
            assert_same(a.find_iter(ac.Input(buf[mis:]).range(s0, s1), as_numpy=True), o.find_iter(hay, span=(s0, s1), as_numpy=True),
                        f"find_iter {mk} mis={mis} ({s0},{s1})")
            w = o.find(hay, span=(s0, s1))
            g = a.find(ac.Input(buf[mis:]).range(s0, s1))
            assert (g is None) == (w is None) and (g is None or (g.pattern(), g.start(), g.end()) == tuple(w))
            calls += 2
a, o = build_pair(pats, "leftmost_first", {"kind": "dfa"})
repl = [bytes([0x41 + (i % 26)]) * (i % 4) for i in range(len(pats))]
got = a.replace_all_bytes(torch.from_numpy(hay).cuda(), repl)
assert bytes(got.cpu().numpy()) == orc.replace_all_bytes(o, hay, repl)
a, o = build_pair(pats, "standard", {"kind": "dfa"})
import io
got = [(m.pattern(), m.start(), m.end()) for m in a.stream_find_iter(io.BytesIO(hay.tobytes()))]
ws = o.find_iter(hay, as_numpy=True)   # StreamChunkIter reports the Standard find_iter sequence (src/automaton.rs:1036-1244)
assert got == list(zip(ws["pattern"].tolist(), ws["start"].tolist(), ws["end"].tolist())), "stream"
calls += 6
torch.cuda.synchronize()
print(f"calls {calls}")
print(f"violations {L.acgpu_guard_violations()}")
