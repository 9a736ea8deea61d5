"""Pins the CPU oracle against every golden vector of the reference's own test-suite
(src/tests.rs:96-642) under the reference's builder-config matrix (src/tests.rs:723-1323)."""
import pytest

import refmatrix
from oracle import orc

MK = {"standard": orc.STANDARD, "leftmost_first": orc.LEFTMOST_FIRST, "leftmost_longest": orc.LEFTMOST_LONGEST}
KIND = {"nnfa": orc.KIND_NNFA, "cnfa": orc.KIND_CNFA, "dfa": orc.KIND_DFA}
SK = {"both": orc.START_BOTH, "anchored": orc.START_ANCHORED, "unanchored": orc.START_UNANCHORED}


def build(pats, mk, kw):
    return orc.Oracle(pats, match_kind=MK[mk], start_kind=SK[kw.get("start_kind", "unanchored")],
                      kind=KIND.get(kw.get("kind"), orc.KIND_AUTO),
                      ascii_case_insensitive=kw.get("ascii_case_insensitive", False),
                      byte_classes=kw.get("byte_classes", True), prefilter=kw.get("prefilter", True),
                      dense_depth=kw.get("dense_depth"))


CASES = list(refmatrix.all_cases())


@pytest.mark.parametrize("cid,mk,api,kw,vectors", CASES, ids=[c[0] for c in CASES])
def test_reference_vectors(cid, mk, api, kw, vectors):
    assert vectors
    for v in vectors:
        pats, hay, want = refmatrix.unhex(v)
        ac = build(pats, mk, kw)
        if api == "find_iter":
            got = ac.find_iter(hay)
        elif api == "overlapping":
            got = ac.find_overlapping_iter(hay)
        else:
            got = ac.find_iter(hay, anchored=True)
        assert got == want, (cid, v["name"], pats, hay)


def test_doctest_vectors():
    for v in refmatrix.doctests():
        pats, hay, want = refmatrix.unhex(v)
        for kind in (orc.KIND_AUTO, orc.KIND_NNFA, orc.KIND_CNFA, orc.KIND_DFA):
            ac = orc.Oracle(pats, kind=kind, ascii_case_insensitive=v["config"].get("ascii_case_insensitive", False))
            got = ac.find_overlapping_iter(hay) if v["api"] == "find_overlapping_iter" else ac.find_iter(hay)
            assert got == want, v["name"]


# API-contract tests, src/tests.rs:1429-1511
@pytest.mark.parametrize("mk", ["leftmost_first", "leftmost_longest"])
def test_overlapping_not_allowed_leftmost(mk):
    ac = orc.Oracle([], match_kind=MK[mk])
    with pytest.raises(orc.OracleError) as e:
        ac.find_overlapping_iter(b"")
    assert e.value.kind == "UnsupportedOverlapping"


@pytest.mark.parametrize("kind", ["nnfa", "cnfa", "dfa"])
def test_anchored_consistency(kind):
    ac = orc.Oracle([b"foo"], kind=KIND[kind], start_kind=orc.START_UNANCHORED)
    with pytest.raises(orc.OracleError) as e:
        ac.find(b"foo", anchored=True)
    assert e.value.kind == "InvalidInputAnchored"
    ac = orc.Oracle([b"foo"], kind=KIND[kind], start_kind=orc.START_ANCHORED)
    with pytest.raises(orc.OracleError) as e:
        ac.find(b"foo", anchored=False)
    assert e.value.kind == "InvalidInputUnanchored"


def test_prefilter_stays_in_bounds():  # src/tests.rs:1522-1530
    ac = orc.Oracle([b"sam", b"frodo", b"pippin", b"merry", b"gandalf", b"sauron"], match_kind=orc.LEFTMOST_FIRST)
    assert ac.find(b"foo gandalf", span=(0, 10)) is None


def test_regression_casei_no_exponential():  # src/tests.rs:1536-1543
    ac = orc.Oracle(["Tsubaki House-Triple Shot Vol01校花三姐妹".encode()], ascii_case_insensitive=True)
    assert ac.find(b"") is None


def test_regression_rare_byte():  # src/tests.rs:1550-1556
    ac = orc.Oracle([b"ab/j/", b"x/"])
    assert ac.find(b"ab/j/", earliest=True) is not None


def test_find_iter_keeps_input_earliest():
    """FindIter keeps the caller's Input (src/automaton.rs:864-883); every step is try_find on it, whose
    `earliest = is_standard() || input.get_earliest()` (:1266).  On a leftmost automaton the earliest flag therefore
    changes the iterator's output: the first match state entered wins, not the leftmost match."""
    for mk in (orc.LEFTMOST_FIRST, orc.LEFTMOST_LONGEST):
        for kind in (orc.KIND_DFA, orc.KIND_CNFA, orc.KIND_NNFA):
            o = orc.Oracle([b"abcd", b"b", b"cd"], match_kind=mk, kind=kind)
            assert o.find_iter(b"abcdabxcd") == [(0, 0, 4), (1, 5, 6), (2, 7, 9)]
            assert o.find_iter(b"abcdabxcd", earliest=True) == [(1, 1, 2), (2, 2, 4), (1, 5, 6), (2, 7, 9)]
    o = orc.Oracle([b"abcd", b"b"])   # Standard: earliest is implied, the flag changes nothing
    assert o.find_iter(b"abcd", earliest=True) == o.find_iter(b"abcd") == [(1, 1, 2)]


def test_regression_case_insensitive_prefilter():  # src/tests.rs:1558-1581
    for c in range(ord("a"), ord("z")):
        for c2 in range(ord("a"), ord("z")):
            needle = bytes([c, c2])
            ac = orc.Oracle([needle], ascii_case_insensitive=True)
            assert len(ac.find_iter(needle.upper())) == 1


def test_chunk_parallel_count_equals_sequential():
    """The seam rule of the multi-threaded CPU baseline (bench.py's all-cores figure): chunk counts add up."""
    import numpy as np
    from oracle import orc
    pats = orc.gen_patterns(300, seed=0xAC01, lo=0x61, span=4)
    pats = [p[: 2 + i % 7] for i, p in enumerate(pats)]
    hay = orc.gen_haystack(0, 1 << 18, seed=0xAC02, lo=0x61, span=4)
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    want, _ = o.dfa_overlapping_count(hay)
    assert want > 10000
    for t in (1, 2, 3, 7, 64, 1000):
        assert o.dfa_overlapping_count_parallel(hay, t) == want
    assert o.dfa_overlapping_count_parallel(hay, 5, span=(1234, 200001)) == o.dfa_overlapping_count(hay, span=(1234, 200001))[0]
