"""not-gpu: the prefix-filter kernels' tables (Bloom tables, exact level-2 maps, trie-only transition table), built by
the host code the upload uses (csrc/host/pf_tables.cpp) and replayed by a CPU model of the kernels' decisions
(acgpu_test_pf_host): the count equals the oracle's overlapping count if and only if the tables let EVERY occurrence
through and level 3 counts each one once.  Kernel 0 = two-type filter (pf_scan.hip), 1 / 2 = large-set filter with its
4-byte / long-prefix level 2 (pfx_scan.hip)."""
import ctypes as C

import numpy as np
import pytest

import aho_corasick_amd as ac
from oracle import orc


def model(pats, hay, kernel, casei=False, kind=ac.AhoCorasickKind.DFA, tails=None):
    b = ac.AhoCorasick.builder().ascii_case_insensitive(casei)
    if tails is not None:
        b.gpu_variant("pfx_tails", tails)
    a = (b.kind(kind) if kind is not None else b).build(pats)
    L = ac.load_test_hooks()
    h = np.ascontiguousarray(hay, dtype=np.uint8)
    n, info = C.c_uint64(), (C.c_uint64 * 8)()
    L.acgpu_test_pf_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    rc = L.acgpu_test_pf_host(a._h, C.c_void_p(h.ctypes.data), len(h), kernel, C.byref(n), info)
    assert rc == 0
    return n.value, dict(pf=int(info[0]), served=int(info[1]) if kernel else int(info[0]), l1=int(info[2]), l2=int(info[3]),
                         depth=int(info[4]), patterns=int(info[5]), exact2=int(info[6]) & 1, fold=(int(info[6]) >> 1) & 1, bits3=int(info[7]) & 1,
                         tail_hits=int(info[6]) >> 8, tail_nodes=int(info[7]) >> 8)


def want(pats, hay, casei=False):
    return len(orc.Oracle(pats, kind=orc.KIND_DFA, ascii_case_insensitive=casei).find_overlapping_iter(hay, as_numpy=True))


def planted(pats, n, seed, lo=0x20, span=95, every=997):
    hay = orc.gen_haystack(0, n, seed=seed, lo=lo, span=span)
    rng = np.random.default_rng(seed)
    for at in range(3, n - 64, every):
        p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
        hay[at:at + len(p)] = p
    for p, at in ((pats[0], 0), (pats[-1], n - len(pats[-1]))):   # both ends of the haystack
        hay[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
    return hay


def test_headline_set_two_type_filter():
    pats = orc.gen_patterns(1000, seed=0xAC01)
    hay = planted(pats, 1 << 20, 1)
    n, info = model(pats, hay, 0)
    assert info["pf"] and not info["exact2"] and info["bits3"]     # (the 4-byte bit table comes with the large-set tables: 256+ patterns of 4+ bytes)
    assert n == want(pats, hay) > 1000
    assert info["l1"] < len(hay) // 2 * 0.03 and info["l2"] <= info["l1"]     # the filter filters (planted text included)


@pytest.mark.parametrize("npat,kernel", [(300, 0), (5000, 0), (5000, 1), (12000, 0), (30000, 0), (30000, 1)])
def test_random_sets_every_kernel(npat, kernel):
    """4 096+ patterns switch the 4-byte bit table on, 24 000+ the exact second table; the large-set kernel serves sets of
    256+ patterns with no pattern shorter than 4 bytes."""
    pats = orc.gen_patterns(npat, seed=0xAC05)
    hay = planted(pats, 1 << 19, npat, every=499)
    n, info = model(pats, hay, kernel)
    assert info["served"]
    if kernel == 0:
        assert info["bits3"] and info["exact2"] == (npat > 24000)
    assert n == want(pats, hay) > 500, info


@pytest.mark.parametrize("minlen", [5, 6, 7, 8, 12])
def test_long_prefix_map(minlen):
    rng = np.random.default_rng(minlen)
    base = orc.gen_patterns(2000, seed=0xAC06 + minlen, lo=0x61, span=26)
    pats = []
    for i, p in enumerate(base):   # every fifth pattern shares a 4..7-byte prefix with its predecessor
        body = (p * 4)[: minlen + int(rng.integers(0, 9))]
        if i % 5 == 4 and pats:
            k = int(rng.integers(4, min(8, minlen) + 1))
            body = pats[-1][:k] + body[k:]
        pats.append(bytes(body))
    hay = planted(pats, 1 << 18, minlen, lo=0x61, span=26, every=211)
    hay[1000:1000 + minlen - 1] = np.frombuffer(pats[3][: minlen - 1], dtype=np.uint8)   # a bare prefix
    w = want(pats, hay)
    for kernel in (1, 2):
        n, info = model(pats, hay, kernel)
        assert info["served"] and n == w > 500, (kernel, info)
    assert info["depth"] == min(8, minlen)
    assert info["l2"] <= model(pats, hay, 1)[1]["l2"]     # the longer exact prefix never lets more through
    # the long-key level 1 (the whole 5..8-byte prefix, bytes beyond it masked out of the window) lets every occurrence
    # through and fewer positions than the four-byte key
    n8, info8 = model(pats, hay, 3)
    assert info8["served"] and n8 == w and info8["l1"] <= info["l1"], (info8, info)
    # ... probed at every other position only when every pattern has nine bytes
    assert bool(model(pats, hay, 4)[1]["served"]) == (minlen >= 9)


def test_short_patterns_wildcards_and_case_insensitive():
    """1- to 3-byte patterns fill in every value of the bytes they do not have; ascii_case_insensitive adds the edges of
    both cases; a-z text makes every table busy."""
    rng = np.random.default_rng(9)
    pats = [bytes(rng.integers(0x61, 0x67, size=int(rng.integers(1, 6)), dtype=np.uint8)) for _ in range(60)]
    hay = rng.integers(0x61, 0x67, size=1 << 14, dtype=np.uint8)
    n, info = model(pats, hay, 0)
    assert info["pf"] and n == want(pats, hay) > 10000
    ci = [b"Needle", b"hAy", b"stack", b"NEEDLES", b"x"]
    text = np.frombuffer(b"a needle in a HAYSTACK of NeEdLeS and hay; xX. " * 300, dtype=np.uint8).copy()
    n, info = model(ci, text, 0, casei=True)
    assert n == want(ci, text, casei=True) > 1500
    assert model([b"", b"a"], text, 0)[1]["pf"] == 0              # an empty pattern: not the filters' automaton


@pytest.mark.parametrize("seed", range(12))
def test_random_automata_all_kernels(seed):
    rng = np.random.default_rng(1000 + seed)
    asz = int(rng.choice([3, 8, 26, 95]))
    lo = 0x61 if asz <= 26 else 0x20
    npat = int(rng.choice([5, 60, 300, 2000]))
    minlen = int(rng.choice([1, 2, 4, 5, 7, 9]))
    pats = []
    for _ in range(npat):
        if pats and rng.random() < 0.2:
            b = pats[int(rng.integers(len(pats)))]
            p = b[: int(rng.integers(minlen, len(b) + 1))] + bytes(rng.integers(lo, lo + asz, size=int(rng.integers(0, 3)), dtype=np.uint8))
        else:
            p = bytes(rng.integers(lo, lo + asz, size=int(rng.integers(minlen, minlen + 8)), dtype=np.uint8))
        pats.append(p)
    n = 1 << 15
    hay = rng.integers(lo, lo + asz, size=n, dtype=np.uint8)
    for at in range(5, n - 32, 257):
        p = np.frombuffer(pats[int(rng.integers(npat))], dtype=np.uint8)
        hay[at:at + len(p)] = p
    w = want(pats, hay)
    for kernel in (0, 1, 2, 3, 4):   # (4: the 8-byte level 1 probed at every other position; sets of 9-byte patterns and longer)
        got, info = model(pats, hay, kernel)
        if info["served"]:
            assert got == w, (seed, kernel, info)
        else:
            assert kernel > 0 and (min(map(len, pats)) < 4 or npat < 256 or (kernel == 3 and min(map(len, pats)) < 8) or
                                   (kernel == 4 and min(map(len, pats)) < 9))


@pytest.mark.parametrize("words", ["words-100", "words-5000", "dictionary-15"])
def test_reference_corpora_natural_text(words):
    """The reference's own benchmark inputs (tests/golden/corpora): English prose against its word lists -- the input on
    which a 4-byte prefix filter is busiest (7 % of the positions are true 4-byte prefixes of words-5000)."""
    import corpora
    pats = corpora.words(words)
    hay = corpora.haystack("sherlock.txt")
    w = len(orc.Oracle(pats, kind=orc.KIND_CNFA).find_overlapping_iter(hay, as_numpy=True))
    assert w >= 10
    for kernel in (0, 1, 2):
        n, info = model(pats, hay, kernel, kind=None)
        if info["served"]:
            assert n == w, (words, kernel, info)
    n4, i4 = model(pats, hay, 1, kind=None)
    n8, i8 = model(pats, hay, 2, kind=None)
    if i4["served"] and i8["depth"] > 4:
        assert i8["l2"] * 3 < i4["l2"]      # the long exact prefix removes most of level 3's work on natural text
    # below most 8-byte prefixes of a dictionary the trie is a chain down to one leaf: those hits are decided by a masked
    # compare with the chain's tail record, not by a walk of dependent trie-row gathers
    if i8["served"] and i8["depth"] == 8:
        assert i8["tail_nodes"] > 0.3 * len(pats), (words, i8)
        if words == "words-5000":
            assert i8["tail_hits"] > 0.7 * i8["l2"], (words, i8)
    # ... and the eight-byte level 1 removes most of level 2's: what survives is little more than the true 8-byte prefixes
    nk, ik = model(pats, hay, 3, kind=None)
    if i8["served"] and i8["depth"] == 8:
        assert ik["served"] and nk == w, (words, ik)
        assert ik["l1"] * 5 < i8["l1"] and ik["l1"] >= ik["l2"] == i8["l2"], (words, ik, i8)
        # ... and probed at every other position (one hash for two starts; every one of these word lists has >= 9-letter words
        # only): the entries are 9-byte prefixes, so fewer starts reach level 2 than true 8-byte prefixes exist
        nx, ix = model(pats, hay, 4, kind=None)
        assert ix["served"] and nx == w, (words, ix)
        assert ix["l2"] <= ik["l2"] and ix["l1"] < 3 * ik["l1"], (words, ix, ik)


@pytest.mark.parametrize("seed", range(8))
def test_key8_every_other_position_random(seed):
    """the 8-byte level 1 probed at the odd offsets only (kernel 4): sets of 9- to 30-byte patterns over 2..26 letters --
    patterns that share 8-byte prefixes and differ in the ninth byte, patterns that start at even and odd positions, at
    position 0 (no probe stands in front of it) and at the very end of the haystack"""
    rng = np.random.default_rng(7700 + seed)
    asz = int(rng.choice([2, 3, 26]))
    pats = []
    for _ in range(int(rng.choice([300, 1500]))):
        if pats and rng.random() < 0.4:
            b = pats[int(rng.integers(len(pats)))]
            k = int(rng.integers(8, len(b) + 1))
            p = b[:k] + bytes(rng.integers(0x61, 0x61 + asz, size=int(rng.integers(max(0, 9 - k), 12)), dtype=np.uint8))
        else:
            p = bytes(rng.integers(0x61, 0x61 + asz, size=int(rng.integers(9, 31)), dtype=np.uint8))
        pats.append(p)
    hay = rng.integers(0x61, 0x61 + asz, size=(1 << 15) + int(rng.integers(0, 3)), dtype=np.uint8)
    for at in range(0, len(hay) - 40, int(rng.choice([58, 61]))):   # (even and odd starts alike)
        p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
        hay[at:at + len(p)] = p
    p = np.frombuffer(pats[1], dtype=np.uint8)
    hay[len(hay) - len(p):] = p
    w = want(pats, hay)
    n3, i3 = model(pats, hay, 3)
    n4, i4 = model(pats, hay, 4)
    assert i3["served"] and i4["served"] and n3 == n4 == w > 400, (seed, i3, i4, n3, n4, w)
    assert i4["l2"] <= i3["l2"]


def test_case_folded_keys_config5_shape():
    """BASELINE config 5's automaton (1 000 patterns, ascii_case_insensitive): every letter edge exists in both cases, the
    tables are built -- and probed -- with case-folded KEYS (one spelling per key instead of up to eight); bit selectors and
    level 3 are untouched.  Mixed-case occurrences must all come through, and the folded tables let far fewer positions of
    random text past level 1 than the 2^k spellings would."""
    pats = orc.gen_patterns(1000, seed=0xAC01)
    hay = orc.gen_haystack(0, 1 << 20, seed=0xAC02)
    for k, pos in enumerate(range(700, len(hay) - 64, 9973)):
        p = pats[k % len(pats)]
        q = p.swapcase() if k % 3 else p.upper()
        hay[pos:pos + len(q)] = np.frombuffer(q, dtype=np.uint8)
    n, info = model(pats, hay, 0, casei=True)
    assert info["pf"] and info["fold"] == 1
    assert n == want(pats, hay, casei=True) > 100
    # the case-sensitive twin of the same set does not fold, and the folded casei tables pass about as few probes as it does
    n_cs, info_cs = model(pats, hay, 0)
    assert info_cs["fold"] == 0
    assert info["l1"] < 3 * max(info_cs["l1"], 1), (info["l1"], info_cs["l1"])


@pytest.mark.parametrize("seed", range(6))
def test_case_folded_keys_random(seed):
    rng = np.random.default_rng(4200 + seed)
    letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 -_@[]{}`~", dtype=np.uint8)
    pats = [bytes(rng.choice(letters, size=int(rng.integers(1, 9)))) for _ in range(int(rng.choice([5, 80, 600])))]
    hay = rng.choice(letters, size=1 << 15).astype(np.uint8)
    for at in range(3, len(hay) - 16, 131):
        p = np.frombuffer(pats[int(rng.integers(len(pats)))].swapcase(), dtype=np.uint8)
        hay[at:at + len(p)] = p
    for casei in (True, False):
        n, info = model(pats, hay, 0, casei=casei)
        assert info["pf"] and n == want(pats, hay, casei=casei), (seed, casei, info)


@pytest.mark.parametrize("tails", [2, 1])
@pytest.mark.parametrize("seed", range(8))
def test_chain_tails_random(seed, tails):
    """long-prefix level 2 with tail records (2: one record per pattern end of a small subtree; 1: chains only): sets of 8- to
    30-byte patterns with shared prefixes, words that end inside another word's tail, duplicates and tails longer than a
    record holds; the tail compares must count exactly what the walk would"""
    rng = np.random.default_rng(9100 + seed)
    asz = int(rng.choice([2, 3, 26]))
    pats = []
    for _ in range(int(rng.choice([300, 1200]))):
        if pats and rng.random() < 0.35:
            b = pats[int(rng.integers(len(pats)))]
            k = int(rng.integers(8, len(b) + 1))
            p = b[:k] + bytes(rng.integers(0x61, 0x61 + asz, size=int(rng.integers(0, 12)), dtype=np.uint8))
        else:
            p = bytes(rng.integers(0x61, 0x61 + asz, size=int(rng.integers(8, 31)), dtype=np.uint8))
        pats.append(p)
    hay = rng.integers(0x61, 0x61 + asz, size=1 << 15, dtype=np.uint8)
    for at in range(3, len(hay) - 40, 61):
        p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
        hay[at:at + len(p)] = p
    w = want(pats, hay)
    hits = {}
    for t in (tails, 0):
        for kernel in (2, 3):
            n, info = model(pats, hay, kernel, tails=t)
            assert info["served"] and info["depth"] == 8 and n == w, (seed, kernel, t, info, n, w)
        hits[t] = info["tail_hits"]
        assert (info["tail_nodes"] > 0 and info["tail_hits"] > 0) if t else info["tail_hits"] == 0
    if tails == 2:   # the records of small subtrees decide more level-2 hits than the chains alone
        _, i1 = model(pats, hay, 3, tails=1)
        assert hits[2] >= i1["tail_hits"]


@pytest.mark.parametrize("seed", range(10))
def test_short_mode_stragglers_beside_long_patterns(seed):
    """Short mode (host/pf_tables.hpp): 300-1500 patterns of >= 9 bytes and one or two stragglers of 3..8 bytes -- the long-key
    tables hold the long patterns only (prefix depth 8, level 1 at every other position), the stragglers are compared at every
    position; stragglers that are prefixes of long patterns, an 8-byte straggler that IS the prefix node of long patterns
    (reported once, not twice), duplicates of a straggler, a straggler at both ends of the haystack.  Variant pfx_short = 0:
    the set as a whole, the same count."""
    rng = np.random.default_rng(7700 + seed)
    asz = int(rng.choice([3, 26]))
    longs = [bytes(rng.integers(0x61, 0x61 + asz, size=int(rng.integers(9, 24)), dtype=np.uint8)) for _ in range(int(rng.choice([300, 1500])))]
    shorts = []
    for _ in range(int(rng.integers(1, 3))):
        k = int(rng.integers(3, 9))
        shorts.append(longs[int(rng.integers(len(longs)))][:k] if rng.random() < 0.6 else bytes(rng.integers(0x61, 0x61 + asz, size=k, dtype=np.uint8)))
    if seed % 3 == 0:
        shorts[0] = longs[0][:8]                      # the prefix node of a long pattern
    pats = longs + shorts + ([shorts[0]] if seed % 2 else [])   # (a duplicate: two ids in one node)
    hay = rng.integers(0x61, 0x61 + asz, size=1 << 15, dtype=np.uint8)
    for at in range(3, len(hay) - 40, 53):
        p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
        hay[at:at + len(p)] = p
    for sp, at in ((shorts[0], 0), (shorts[-1], len(hay) - len(shorts[-1]))):
        hay[at:at + len(sp)] = np.frombuffer(sp, dtype=np.uint8)
    w = want(pats, hay)
    n4, i4 = model(pats, hay, 4)
    distinct = len(set(shorts))
    assert i4["served"] and i4["depth"] == 8 and n4 == w, (seed, shorts, i4, n4, w)
    for kernel in (2, 3):     # the other long-key kernels do not run on these tables
        assert not model(pats, hay, kernel)[1]["served"], (seed, kernel)
    n1, i1 = model(pats, hay, 1)
    assert bool(i1["served"]) == (min(map(len, shorts)) >= 4) and (not i1["served"] or n1 == w), (seed, shorts, i1)
    assert model(pats, hay, 0)[0] == w            # the two-type filter knows every pattern
    b = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_variant("pfx_short", 0)
    a0 = b.build(pats)
    L = ac.load_test_hooks()
    nn, info = C.c_uint64(), (C.c_uint64 * 8)()
    hh = np.ascontiguousarray(hay)
    for kernel in (1, 2, 3):
        if L.acgpu_test_pf_host(a0._h, C.c_void_p(hh.ctypes.data), len(hh), kernel, C.byref(nn), info) == 0 and int(info[1]):
            assert nn.value == w, (seed, kernel, "pfx_short=0")
    assert distinct in (1, 2)
