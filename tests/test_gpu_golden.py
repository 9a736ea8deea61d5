"""-m gpu: every golden vector of the reference (src/tests.rs:96-642) under the reference's
builder-config matrix (src/tests.rs:723-1323), through the C ABI / HIP path."""
import pytest
import torch

import aho_corasick_amd as ac
import refmatrix
from gpu_util import build_pair

pytestmark = pytest.mark.gpu
CASES = list(refmatrix.all_cases())


@pytest.mark.parametrize("engine", ["auto", "walk"])
@pytest.mark.parametrize("cid,mk,api,kw,vectors", CASES, ids=[c[0] for c in CASES])
def test_reference_vectors_gpu(cid, mk, api, kw, vectors, engine):
    """engine "auto": the fastest device path (prefix filter / LDS rows / derived full DFA for NFA kinds, parallel
    find_iter, windowed find); "walk": only the reference-faithful engines (the automaton's own DFA or the
    contiguous-NFA failure-link walk, one-lane FindIter)."""
    for v in vectors:
        pats, hay, want = refmatrix.unhex(v)
        a, _ = build_pair(pats, mk, kw, engine=engine)
        if api == "find_iter":
            got = [m.as_tuple() for m in a.find_iter(hay)]
        elif api == "overlapping":
            got = [m.as_tuple() for m in a.find_overlapping_iter(hay)]
        else:
            got = [m.as_tuple() for m in a.try_find_iter(ac.Input(hay).anchored(ac.Anchored.Yes))]
        assert got == want, (cid, v["name"], pats, hay)


@pytest.mark.parametrize("engine", ["walk", "hot", "pf"])
@pytest.mark.parametrize("shift", [0, 1, 50, 57, 63, 120])
def test_overlapping_vectors_across_chunk_seams(engine, shift):
    """Same vectors searched as a sub-span [shift, shift+len) of a larger device buffer with 64-byte
    lane-chunks, so the haystack straddles chunk seams; offsets are absolute (src/util/search.rs:41-48)."""
    vectors = refmatrix.collection("AC_STANDARD_OVERLAPPING")
    for v in vectors:
        pats, hay, want = refmatrix.unhex(v)
        a, _ = build_pair(pats, "standard", {"kind": "dfa"}, chunk=64, engine=engine)
        full = bytearray(b"\x7f" * 256)
        full[shift:shift + len(hay)] = hay
        t = torch.frombuffer(full, dtype=torch.uint8).cuda()
        try:
            got = [m.as_tuple() for m in a.find_overlapping_iter(ac.Input(t).range(shift, shift + len(hay)))]
        except RuntimeError as e:  # no prefix filter for pattern sets that are empty or hold an empty pattern
            assert engine == "pf" and "invalid argument" in str(e) and (not pats or any(len(p) == 0 for p in pats))
            continue
        assert got == [(p, s + shift, e + shift) for p, s, e in want], (v["name"], shift, engine)


def test_doctest_vectors_gpu():
    for v in refmatrix.doctests():
        pats, hay, want = refmatrix.unhex(v)
        for kind in (None, "nnfa", "cnfa", "dfa"):
            a, _ = build_pair(pats, "standard", {"kind": kind,
                                                "ascii_case_insensitive": v["config"].get("ascii_case_insensitive", False)})
            it = a.find_overlapping_iter(hay) if v["api"] == "find_overlapping_iter" else a.find_iter(hay)
            assert [m.as_tuple() for m in it] == want, v["name"]


def test_find_and_is_match_gpu():  # src/tests.rs:1522-1556 + README
    a = ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).build(
        [b"sam", b"frodo", b"pippin", b"merry", b"gandalf", b"sauron"])
    assert a.find(ac.Input(b"foo gandalf").range(0, 10)) is None
    assert a.find(b"foo gandalf") == (4, 4, 11)
    b = ac.AhoCorasick.new([b"ab/j/", b"x/"])
    assert b.is_match(b"ab/j/")
    assert not b.is_match(b"ab/j")
    c = ac.AhoCorasick.new([b"foo", b"bar", b"quux", b"baz"])
    assert c.is_match(b"xxx bar xxx") and not c.is_match(b"xxx qux xxx")
