"""-m gpu: short mode of the large-set filter (csrc/host/pf_tables.hpp, device/pfx_scan.hip): a dictionary of patterns of nine
bytes and more with one or two stragglers of 3..8 bytes is searched in ONE pass -- the long-key tables hold the long patterns,
the producer wavefronts compare the stragglers' first bytes at every position, the verifiers all of them -- and reports what
the reference reports for the whole set (/root/reference/src/automaton.rs:1021-1053), in its order."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
import corpora
from gpu_util import assert_same
from oracle import orc

pytestmark = pytest.mark.gpu

EXTRA = {8: b"daughter", 7: b"morning", 6: b"window", 5: b"chair", 4: b"lamp", 3: b"cab"}


def prose(n):
    text = corpora.haystack("sherlock.txt")
    return np.tile(text, -(-n // len(text)))[:n].copy()


@pytest.mark.parametrize("k", [8, 7, 6, 5, 4, 3])
def test_dictionary_with_one_straggler(k):
    words = list(corpora.words("words-5000")) + [EXTRA[k]]
    hay = prose((24 << 20) + 12345)
    tail = np.frombuffer(EXTRA[k], dtype=np.uint8)
    hay[:k] = tail                      # the scan's first start position
    hay[len(hay) - k:] = tail           # ... and an occurrence that ends with the haystack
    o = orc.Oracle(words, kind=orc.KIND_DFA)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    d = torch.from_numpy(hay).cuda()
    a = ac.AhoCorasick.builder().build(words)
    prof = ac._lib.CProfile()
    got = a.find_overlapping_iter(d, as_numpy=True, profile=prof)
    assert int(prof.engine_used) == 4, int(prof.engine_used)   # the filter engine on the whole set: no split into two automata
    assert_same(got, want, f"straggler of {k} bytes, host output")
    out = torch.zeros(len(want) * 24 + 24, dtype=torch.uint8, device="cuda")
    for call in range(3):               # (the second and third through the fused order chain)
        m, ok = a.overlapping_device(d, out=out)
        assert ok and m == len(want)
        assert_same(out[: m * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, f"straggler of {k} bytes, device call {call}")
    m, ok = a.overlapping_device(d, out=None)       # count only: chunk counters instead of events
    assert m == len(want)
    # spans that end inside / right behind an occurrence of the straggler at the end of the haystack
    n = len(hay)
    for end in (n - 1, n - k + 2 if k > 2 else n, n):
        w = o.find_overlapping_iter(hay, span=(5, end), as_numpy=True)
        assert_same(a.find_overlapping_iter(ac.Input(d).range(5, end), as_numpy=True), w, f"span (5, {end})")
    # the set as a whole (variant pfx_short = 0) and find_iter over the same stream
    a0 = ac.AhoCorasick.builder().gpu_variant("pfx_short", 0).build(words)
    assert_same(a0.find_overlapping_iter(d, as_numpy=True), want, "pfx_short = 0")
    lf = ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).build(words)
    olf = orc.Oracle(words, match_kind=1, kind=orc.KIND_DFA)
    assert_same(lf.find_iter(d, as_numpy=True), olf.find_iter(hay, as_numpy=True), "LeftmostFirst find_iter")


def test_two_stragglers_prefixes_of_long_words_and_duplicates():
    base = list(corpora.words("words-5000"))
    eight = next(w[:8] for w in base if len(w) > 10)            # the prefix node of at least one long word
    words = base + [eight, b"hol", eight]                        # ... twice: two pattern ids in one node
    hay = prose(16 << 20)
    o = orc.Oracle(words, kind=orc.KIND_DFA)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    d = torch.from_numpy(hay).cuda()
    a = ac.AhoCorasick.builder().build(words)
    prof = ac._lib.CProfile()
    assert_same(a.find_overlapping_iter(d, as_numpy=True, profile=prof), want, "two stragglers")
    assert int(prof.engine_used) == 4
    three = base + [b"cab", b"lamp", b"chair"]                   # three distinct stragglers: not short mode (split set), same answer
    o3 = orc.Oracle(three, kind=orc.KIND_DFA)
    a3 = ac.AhoCorasick.builder().build(three)
    assert_same(a3.find_overlapping_iter(d, as_numpy=True), o3.find_overlapping_iter(hay, as_numpy=True), "three stragglers")


def test_random_sets_in_short_mode():
    rng = np.random.default_rng(0x5407)
    for case in range(6):
        asz = int(rng.choice([4, 26]))
        longs = [bytes(rng.integers(0x61, 0x61 + asz, size=int(rng.integers(9, 20)), dtype=np.uint8)) for _ in range(1200)]
        shorts = [longs[int(rng.integers(len(longs)))][: int(rng.integers(3, 9))] for _ in range(int(rng.integers(1, 3)))]
        pats = longs + shorts
        n = int(rng.choice([1 << 16, (1 << 20) + 7, 3 << 20]))
        hay = rng.integers(0x61, 0x61 + asz, size=n, dtype=np.uint8)
        for at in range(5, n - 40, 997):
            p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
            hay[at:at + len(p)] = p
        o = orc.Oracle(pats, kind=orc.KIND_DFA)
        a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_engine("pf").gpu_variant("pfx_min_patterns", 1).build(pats)
        d = torch.from_numpy(hay).cuda()
        s0 = int(rng.integers(0, 100)); s1 = n - int(rng.integers(0, 9))
        assert_same(a.find_overlapping_iter(d, as_numpy=True), o.find_overlapping_iter(hay, as_numpy=True), f"case {case}")
        assert_same(a.find_overlapping_iter(ac.Input(d).range(s0, s1), as_numpy=True), o.find_overlapping_iter(hay, span=(s0, s1), as_numpy=True),
                    f"case {case} span ({s0},{s1})")
