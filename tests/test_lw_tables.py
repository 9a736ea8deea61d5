"""LDS-walk engine (device/lds_walk.hip) without a GPU: its host tables (dense rows, single-exception handles, exception
chains -- host/lw_tables.cpp) walked by the CPU emulation of the kernel's step rules (fast step, per-dword flag, exact
redo) must count exactly what the oracle's overlapping search finds (acgpu_test_lw_host is a test hook, not a search
path)."""
import ctypes as C

import numpy as np
import pytest

import aho_corasick_amd as ac
from oracle import orc


FLAVOURS = {None: 0, "narrow": 1, "wide": 2, "full": 3}
CLS = {None: 0, "lds": 1, "computed": 2}


def lw(pats, hay, flavour=None, cls=None, **kw):
    b = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA)
    if kw.get("casei"):
        b.ascii_case_insensitive(True)
    if kw.get("byte_classes") is False:
        b.byte_classes(False)
    a = b.build(pats)
    L = ac.load_test_hooks()
    n, info = C.c_uint64(), (C.c_uint64 * 8)()
    info[0], info[1] = FLAVOURS[flavour], CLS[cls]
    h = np.ascontiguousarray(hay)
    rc = L.acgpu_test_lw_host(a._h, C.c_void_p(h.ctypes.data), len(h), C.byref(n), info)
    assert rc == 0
    return n.value, dict(eligible=int(info[0]), image=int(info[1]), dense=int(info[2]), multi=int(info[3]),
                         classes=int(info[4]), states=int(info[5]), redo=int(info[6]), wide=int(info[7]) & 1, full=(int(info[7]) >> 1) & 1,
                         computed=(int(info[7]) >> 2) & 1, monotone=(int(info[7]) >> 3) & 1, disjoint=(int(info[7]) >> 4) & 1,
                         redo_est_ppm=int(info[7]) >> 8)


def want(pats, hay, **kw):
    o = orc.Oracle(pats, kind=orc.KIND_DFA, ascii_case_insensitive=bool(kw.get("casei")),
                   byte_classes=kw.get("byte_classes", True))
    return len(o.find_overlapping_iter(hay, as_numpy=True))


def test_headline_automaton_layout_and_counts():
    pats = orc.gen_patterns(1000, seed=0xAC01)
    hay = orc.gen_haystack(0, 1 << 21, seed=0xAC02)
    for k, pos in enumerate(range(4090, len(hay) - 64, 65521)):
        p = np.frombuffer(pats[k % len(pats)], dtype=np.uint8)
        hay[pos:pos + len(p)] = p
    w = want(pats, hay)
    for cls, ncls in (("lds", 96), ("computed", 97)):            # computed: 0x20..0x7E one class each + "below" + "above"
        n, info = lw(pats, hay, cls=cls)
        assert info["eligible"] and info["states"] == 9289 - 2   # hids: every state but FAIL and the anchored start
        assert info["classes"] == ncls and info["image"] <= 160 * 1024 and not info["full"] and not info["wide"]
        assert info["computed"] == (cls == "computed")
        assert info["dense"] >= 96                               # start state + the 95 states at distance 1
        assert n == w > 30
        # the exact path is the exception (a dword that holds a match no longer takes it: its length comes from a table)
        assert info["redo"] < (len(hay) // 4) // 200, info
    assert lw(pats, hay)[1]["computed"] == 1                     # the engine's own choice for printable-ASCII sets
    assert lw(pats, hay, flavour="full")[1]["eligible"] == 0     # 9 287 rows of 388 bytes do not fit


@pytest.mark.parametrize("seed", range(6))
def test_random_small_alphabets(seed):
    """Dense matches, nested / duplicate / 1-byte patterns, many multi states."""
    rng = np.random.default_rng(500 + seed)
    for case in range(10):
        sigma = int(rng.integers(2, 7))
        pats = [bytes(rng.integers(0x61, 0x61 + sigma, size=int(rng.integers(1, 9)), dtype=np.uint8))
                for _ in range(int(rng.integers(1, 60)))]
        hay = rng.integers(0x61, 0x61 + sigma + 1, size=int(rng.integers(0, 5000)), dtype=np.uint8)
        w = want(pats, hay)
        n, info = lw(pats, hay)
        assert info["eligible"] and info["full"]                 # a few dozen states: one row each
        assert n == w, (seed, case, info)
        for flavour in ("narrow", "wide", "full"):
            for cls in ("lds", "computed"):
                n, info = lw(pats, hay, flavour=flavour, cls=cls)
                assert info["eligible"] and n == w, (seed, case, flavour, cls, info)


def test_case_insensitive_classes_merge():
    """Both cases of a letter share every DFA column: the engine's own class map merges them, so a trie edge stays ONE
    exception (with the reference's classes it would be two and every deep state a multi state)."""
    pats = orc.gen_patterns(1000, seed=0xAC01)
    hay = orc.gen_haystack(0, 1 << 19, seed=0xAC07)
    for k, pos in enumerate(range(100, len(hay) - 100, 2999)):
        p = np.frombuffer(pats[k % len(pats)].swapcase(), dtype=np.uint8)
        hay[pos:pos + len(p)] = p
    n, info = lw(pats, hay, casei=True)
    assert info["eligible"] and info["classes"] <= 96 - 26 + 1 and not info["computed"]   # merged classes are no clamp of the byte
    assert lw(pats, hay, casei=True, cls="computed")[1]["eligible"] == 0
    assert n == want(pats, hay, casei=True) > 100
    n2, info2 = lw(pats, hay, casei=True, byte_classes=False)     # 256 reference classes: same engine classes
    assert info2["classes"] == info["classes"] and n2 == n


def test_words_with_shared_prefixes_and_long_chains():
    """Dictionary-like sets: depth-2+ states with many children become dense rows while LDS lasts, the rest chain
    their exceptions; all 256 byte values in the haystack."""
    rng = np.random.default_rng(9)
    stems = [b"inter", b"intra", b"intro", b"in", b"int", b"un", b"under", b"over", b"re", b"pre", b"pro", b"con"]
    tails = [b"", b"s", b"ed", b"ing", b"ion", b"ions", b"al", b"ally", b"ness", b"ment", b"able", b"ible", b"er", b"est"]
    mids = [b"act", b"ect", b"uct", b"form", b"port", b"press", b"sist", b"tain", b"vent", b"view", b"duce", b"fer"]
    pats = sorted({s + m + t for s in stems for m in mids for t in tails})
    text = b" ".join(pats[int(i)] for i in rng.integers(0, len(pats), size=4000))
    hay = np.frombuffer(text, dtype=np.uint8).copy()
    noise = rng.integers(0, 256, size=len(hay), dtype=np.uint8)
    hay = np.where(rng.random(len(hay)) < 0.05, noise, hay).astype(np.uint8)
    w = want(pats, hay)
    for flavour in (None, "narrow", "wide", "full"):
        for cls in (None, "lds", "computed"):
            n, info = lw(pats, hay, flavour=flavour, cls=cls)
            if flavour == "full":
                assert not info["eligible"]                      # ~5 000 states
                continue
            assert info["eligible"] and info["multi"] >= 0, (flavour, cls)
            assert n == w > 4000, (flavour, cls, info)


def test_too_large_for_lds_is_refused():
    pats = orc.gen_patterns(12000, seed=0xAC05)
    n, info = lw(pats, np.zeros(16, dtype=np.uint8))
    assert info["eligible"] == 0 and n == 0


def test_empty_pattern_every_state_matches():
    pats = [b"", b"a", b"ba"]
    hay = np.frombuffer(b"abbaababbab" * 30, dtype=np.uint8).copy()
    for flavour in (None, "narrow", "full"):
        for cls in ("lds", "computed"):
            n, info = lw(pats, hay, flavour=flavour, cls=cls)
            assert info["eligible"] and n == want(pats, hay), (flavour, cls)


def test_reference_corpora_natural_text():
    """English prose against the reference's words-100 list: the automaton fits the engine (the GPU's automatic choice
    hands match-dense inputs of such sets to it)."""
    import corpora
    pats = corpora.words("words-100")
    hay = corpora.haystack("sherlock.txt")
    w = want(pats, hay)
    for cls in ("lds", "computed"):
        n, info = lw(pats, hay, cls=cls)
        assert info["eligible"] and n == w >= 10, (cls, info)


def test_small_alphabet_gets_the_wide_base_layout():
    """1 000 a-z patterns: 27 engine classes, rows of 27 dwords (an odd stride: 108 bytes).  With the 8-bit row index only
    254 of the states that differ from their dense ancestor in two or more columns got a row and the rest became exception
    chains (half of the dwords of an a-z haystack took the exact path); the 10 | 6 | 16 handle layout fills LDS with rows
    (772: every state of depth <= 2 and the busiest of depth 3; 625 with the 128-byte rows of rounds 1-3) and 0.3 % of the
    dwords of pattern-like input are left on the exact path (2 % with 625 rows)."""
    pats = orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)
    hay = orc.gen_haystack(0, 1 << 20, seed=0xAC02, lo=0x61, span=26)
    n, info = lw(pats, hay)
    assert info["eligible"] and info["wide"] == 1 and info["classes"] <= 28 and not info["computed"]   # wide: LDS class map
    assert info["dense"] > 500, info
    assert n == want(pats, hay) > 100
    assert info["redo"] < (len(hay) // 4) // 25, info
    # the headline set (96 classes) keeps the narrow layout: its fast step is two VALU operations shorter
    n2, info2 = lw(orc.gen_patterns(1000, seed=0xAC01), orc.gen_haystack(0, 1 << 16, seed=3))
    assert info2["wide"] == 0
    # what the routing rule is told (build_lw_tables): pattern-like input keeps the a-z walk on its exact path in ~0.3 % of
    # the dwords (20 % of the wave-dwords; 2 % / 73 % with the power-of-two rows), the headline set's in well under 0.1 %
    assert 1_000 < info["redo_est_ppm"] < 8_000, info
    assert info2["redo_est_ppm"] < 8_000, info2


@pytest.mark.parametrize("seed", range(4))
def test_wide_layout_random_sets(seed):
    """Sets large enough to want more than 254 rows over alphabets of at most 64 classes, with nested and duplicate patterns."""
    rng = np.random.default_rng(900 + seed)
    sigma = int(rng.integers(8, 40))
    pats = [bytes(rng.integers(0x41, 0x41 + sigma, size=int(rng.integers(2, 7)), dtype=np.uint8)) for _ in range(1500)]
    hay = rng.integers(0x41, 0x41 + sigma + 1, size=200_000, dtype=np.uint8)
    w = want(pats, hay)
    for cls in ("lds", "computed"):
        n, info = lw(pats, hay, cls=cls)
        assert info["eligible"], info
        assert n == w
        assert info["wide"] == 1 or info["dense"] <= 254, info


@pytest.mark.parametrize("seed", range(4))
def test_full_flavour_match_dense_small_sets(seed):
    """The reference's small-set definitions (teddy / memchr / same families): one row per state, every byte may end
    several patterns, lengths come out of the handles; all 256 byte values, patterns of bytes far apart (wide clamp)."""
    rng = np.random.default_rng(7000 + seed)
    for case in range(8):
        npat = int(rng.integers(1, 40))
        alphabet = rng.choice(256, size=int(rng.integers(1, 12)), replace=False).astype(np.uint8)
        pats = [bytes(rng.choice(alphabet, size=int(rng.integers(1, 5)))) for _ in range(npat)]
        hay = np.where(rng.random(6000) < 0.7, rng.choice(alphabet, size=6000), rng.integers(0, 256, size=6000)).astype(np.uint8)
        w = want(pats, hay)
        n, info = lw(pats, hay)
        assert info["eligible"] and info["full"] and n == w, (seed, case, info)
        for cls in ("lds", "computed"):
            n, info = lw(pats, hay, flavour="full", cls=cls)
            assert (not info["eligible"] and cls == "computed") or n == w, (seed, case, cls, info)


def test_match_list_lengths_beyond_one():
    """Nested patterns: one state ends several patterns; duplicates count twice (src/dfa.rs:275-279)."""
    pats = [b"a", b"aa", b"aaa", b"a", b"ba", b"aba", b"a" * 9]
    hay = np.frombuffer(b"aaaaabaaabaaaaaaaaaaaaab" * 50, dtype=np.uint8).copy()
    w = want(pats, hay)
    for flavour in ("narrow", "full"):
        for cls in ("lds", "computed"):
            n, info = lw(pats, hay, flavour=flavour, cls=cls)
            assert info["eligible"] and n == w, (flavour, cls, n, w)


def lw_records(pats, hay, **kw):
    b = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA)
    if kw.get("casei"):
        b.ascii_case_insensitive(True)
    a = b.build(pats)
    L = ac.load_test_hooks()
    h = np.ascontiguousarray(hay)
    cap = 8 * len(h) + 64
    out = np.zeros(cap, dtype=ac.MATCH_DTYPE)
    n, served = C.c_size_t(), C.c_int32()
    rc = L.acgpu_test_lw_records_host(a._h, C.c_void_p(h.ctypes.data), len(h), C.c_void_p(out.ctypes.data), cap, C.byref(n), C.byref(served))
    assert rc == 0, rc
    return (out[: n.value] if served.value else None)


@pytest.mark.parametrize("casei", [False, True])
def test_full_flavour_record_lists(casei):
    """The match lists in the LDS image of the one-row-per-state form (what k_lw_fill writes records from): pattern ids in
    the reference's match-list order (src/dfa.rs:275-279), nested / duplicate / empty patterns, every byte value."""
    rng = np.random.default_rng(77)
    for case in range(12):
        alphabet = rng.choice(256, size=int(rng.integers(2, 8)), replace=False).astype(np.uint8)
        if casei:
            alphabet = np.array([0x61, 0x42, 0x63, 0x44, 0x7A, 0x20][: len(alphabet)], dtype=np.uint8)
        pats = [bytes(rng.choice(alphabet, size=int(rng.integers(0 if case % 4 == 0 else 1, 6)))) for _ in range(int(rng.integers(1, 25)))]
        pats += pats[:2]
        hay = np.where(rng.random(3000) < 0.8, rng.choice(alphabet, size=3000), rng.integers(0, 256, size=3000)).astype(np.uint8)
        if casei:
            hay = np.where(rng.random(3000) < 0.5, hay ^ 0x20, hay).astype(np.uint8)
        got = lw_records(pats, hay, casei=casei)
        assert got is not None, (case, pats)
        want_rec = orc.Oracle(pats, kind=orc.KIND_DFA, ascii_case_insensitive=casei).find_overlapping_iter(hay, as_numpy=True)
        assert len(got) == len(want_rec), (case, len(got), len(want_rec))
        for f in ("pattern", "start", "end"):
            assert np.array_equal(got[f], want_rec[f]), (case, f)


def lw_event_records(pats, hay, chunk, **kw):
    b = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA)
    if kw.get("casei"):
        b.ascii_case_insensitive(True)
    a = b.build(pats)
    L = ac.load_test_hooks()
    h = np.ascontiguousarray(hay)
    cap = 8 * len(h) + 64
    out = np.zeros(cap, dtype=ac.MATCH_DTYPE)
    n, served = C.c_size_t(), C.c_int32()
    rc = L.acgpu_test_lw_event_records_host(a._h, C.c_void_p(h.ctypes.data), len(h), chunk, C.c_void_p(out.ctypes.data), cap, C.byref(n), C.byref(served))
    assert rc == 0, rc
    return (out[: n.value] if served.value else None)


@pytest.mark.parametrize("chunk", [16, 64, 128, 512])
def test_event_form_host_model(chunk):
    """The event form of the LDS walk (device/lds_emit.hip) modelled on the host: one event per dword that gained a record
    (state before the dword, records of the lane-chunk before it, byte masks on the ragged edges), the scan of the lane-chunk
    counts and the re-walk of the events -- in reverse order of arrival -- give the reference's overlapping stream
    (src/automaton.rs:1021-1053) whatever the lane-chunk: patterns longer than a lane-chunk, nested and duplicate patterns,
    haystacks that are not whole dwords or whole lane-chunks, an empty haystack."""
    rng = np.random.default_rng(1234 + chunk)
    for case in range(10):
        alphabet = rng.choice(256, size=int(rng.integers(2, 6)), replace=False).astype(np.uint8)
        longest = 40 if case % 3 == 0 else 6
        pats = [bytes(rng.choice(alphabet, size=int(rng.integers(1, longest)))) for _ in range(int(rng.integers(1, 20)))]
        pats += pats[:1]
        n = [0, 1, 3, 17, chunk - 1, chunk, chunk + 1, 2999, 3000, 4096][case]
        hay = np.where(rng.random(n) < 0.85, rng.choice(alphabet, size=n), rng.integers(0, 256, size=n)).astype(np.uint8)
        got = lw_event_records(pats, hay, chunk)
        assert got is not None, (case, pats)
        want = orc.Oracle(pats, kind=orc.KIND_DFA).find_overlapping_iter(hay, as_numpy=True)
        assert len(got) == len(want), (case, len(got), len(want))
        for f in ("pattern", "start", "end"):
            assert np.array_equal(got[f], want[f]), (case, f)


def test_full_flavour_records_unavailable_for_large_automata():
    assert lw_records(orc.gen_patterns(1000, seed=0xAC01), np.zeros(16, dtype=np.uint8)) is None


def test_disjoint_sets_iterate_every_occurrence():
    """LwHostTables::disjoint (every match state is a sync state holding one pattern): occurrences of such a set can neither
    overlap nor share an end, so the non-overlapping iteration of EVERY match kind takes every occurrence -- the property
    capi_find.cpp rests on when it serves find_iter by the overlapping search.  Checked on the oracle for random small sets;
    and the flag itself on sets where it is known."""
    hay0 = np.frombuffer(b"x", dtype=np.uint8)
    for pats, dis in (([b"a"], 1), ([b"a", b"b", b"\n"], 1), ([b"the"], 1), ([b"the", b"you"], 1), ([b"aa"], 0), ([b"ab", b"b"], 0),
                      ([b"abc", b"bcd"], 0), ([b"a", b"a"], 0), ([b"abab"], 0), ([b"ab", b"cd"], 1), ([b"ab", b"ba"], 0)):
        assert lw(pats, hay0)[1]["disjoint"] == dis, pats
    rng = np.random.default_rng(17)
    seen = 0
    for trial in range(400):
        alpha = [b"ab", b"abc", b"abcde"][trial % 3]
        pats = [bytes(rng.choice(np.frombuffer(alpha, dtype=np.uint8), size=int(rng.integers(1, 4)))) for _ in range(int(rng.integers(1, 5)))]
        hay = rng.choice(np.frombuffer(alpha + b"x", dtype=np.uint8), size=300).astype(np.uint8)
        info = lw(pats, hay)[1]
        if not (info["eligible"] and info["full"] and info["disjoint"]):
            continue
        seen += 1
        ov = orc.Oracle(pats, kind=orc.KIND_DFA).find_overlapping_iter(hay)
        for mk in (0, 1, 2):
            assert orc.Oracle(pats, match_kind=mk).find_iter(hay) == ov, (pats, mk)
    assert seen > 40


def test_monotone_sets_meet_their_occurrences_in_start_order():
    """LwHostTables::monotone: no pattern is a proper prefix or a proper infix of another (every match state is a leaf of the trie)."""
    hay0 = np.frombuffer(b"x", dtype=np.uint8)
    for pats, mono in (([b"abc", b"bc", b"c"], 1), ([b"ab", b"abc"], 0), ([b"abcd", b"bc"], 0), ([b"the", b"you", b"and"], 1),
                       ([b"a", b"ab"], 0), ([b"xab", b"ab"], 1)):
        assert lw(pats, hay0)[1]["monotone"] == mono, pats
