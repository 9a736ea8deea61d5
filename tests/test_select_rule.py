"""The non-overlapping selection rule of the parallel find_iter (aho-corasick_amd/csrc/device/select.hpp), run on
the HOST through the acgpu_test_select_host test hook: oracle overlapping stream (Standard automaton of the same
patterns) -> rule -> must equal the oracle's find_iter for every match kind.  No GPU needed."""
import ctypes as C
import random

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import aho_corasick_amd as ac
import refmatrix
from oracle import orc

L = ac.load_test_hooks()


def select(stream, mk, span_start, max_len):
    s = np.ascontiguousarray(stream)
    out = np.empty(max(len(s), 1), dtype=ac.MATCH_DTYPE)
    n = C.c_size_t()
    rc = L.acgpu_test_select_host(C.c_void_p(s.ctypes.data), len(s), mk, span_start, max_len,
                                  C.c_void_p(out.ctypes.data), len(out), C.byref(n))
    assert rc == 0
    return [(int(p), int(a), int(b)) for p, a, b in zip(out["pattern"][:n.value], out["start"][:n.value], out["end"][:n.value])]


def check(pats, hay, mk, casei=False, span=None):
    if not pats or any(len(p) == 0 for p in pats):
        return
    occ = orc.Oracle(pats, kind=orc.KIND_DFA, ascii_case_insensitive=casei)
    stream = occ.find_overlapping_iter(hay, span=span, as_numpy=True)
    want = orc.Oracle(pats, match_kind=mk, kind=orc.KIND_DFA, ascii_case_insensitive=casei).find_iter(hay, span=span)
    got = select(stream, mk, 0 if span is None else span[0], occ.max_pattern_len)
    assert got == want, (pats, hay, mk, casei, span)


def test_rule_on_reference_vectors():
    for coll, mk in (("AC_STANDARD_NON_OVERLAPPING", orc.STANDARD), ("AC_LEFTMOST_FIRST", orc.LEFTMOST_FIRST),
                     ("AC_LEFTMOST_LONGEST", orc.LEFTMOST_LONGEST)):
        for v in refmatrix.collection(coll):
            pats, hay, _ = refmatrix.unhex(v)
            check(pats, hay, mk)
    for v in refmatrix.VECTORS["groups"]["ASCII_CASE_INSENSITIVE"] + \
            refmatrix.VECTORS["groups"]["ASCII_CASE_INSENSITIVE_NON_OVERLAPPING"]:
        pats, hay, _ = refmatrix.unhex(v)
        for mk in (orc.STANDARD, orc.LEFTMOST_FIRST, orc.LEFTMOST_LONGEST):
            check(pats, hay, mk, casei=True)


@st.composite
def case(draw):
    alpha = draw(st.sampled_from([b"ab", b"abc", b"aAbB", b"abcd"]))
    pats = draw(st.lists(st.lists(st.sampled_from(list(alpha)), min_size=1, max_size=6).map(bytes), min_size=1,
                         max_size=8))
    hay = bytes(draw(st.lists(st.sampled_from(list(alpha)), min_size=0, max_size=80)))
    return pats, hay


@settings(max_examples=600, deadline=None)
@given(case(), st.sampled_from([orc.STANDARD, orc.LEFTMOST_FIRST, orc.LEFTMOST_LONGEST]), st.booleans(),
       st.integers(0, 10), st.integers(0, 10))
def test_rule_equals_oracle_find_iter(c, mk, casei, cut_a, cut_b):
    pats, hay = c
    check(pats, hay, mk, casei)
    a = min(cut_a, len(hay))
    b = max(a, len(hay) - cut_b)
    check(pats, hay, mk, casei, span=(a, b))


def test_rule_on_synthetic_c5():
    pats = orc.gen_patterns(1000, seed=0xAC01)
    hay = orc.gen_haystack(0, 1 << 20, seed=0xAC05)
    rng = random.Random(1)
    for k in range(400):
        p = pats[rng.randrange(len(pats))].swapcase()
        pos = rng.randrange(0, len(hay) - 32)
        hay[pos:pos + len(p)] = np.frombuffer(p, dtype=np.uint8)
    for mk in (orc.STANDARD, orc.LEFTMOST_FIRST, orc.LEFTMOST_LONGEST):
        check(pats, hay, mk, casei=True)
    # nested / duplicate patterns, dense matches
    pats2 = [b"ab", b"abab", b"b", b"ba", b"abab", b"bab", b"a"]
    hay2 = np.frombuffer(b"abababbbaabab" * 500, dtype=np.uint8).copy()
    for mk in (orc.STANDARD, orc.LEFTMOST_FIRST, orc.LEFTMOST_LONGEST):
        check(pats2, hay2, mk)


def windowed(pats, hay, mk, casei, w):
    """Host model of capi_find.cpp::nonoverlapping_windowed (the device path for occurrence streams that do not fit): per
    window (pos, b] the occurrences ending in it, selection from `pos`, final-match rule, floor advance."""
    occ = orc.Oracle(pats, kind=orc.KIND_DFA, ascii_case_insensitive=casei)
    stream = occ.find_overlapping_iter(hay, as_numpy=True)
    Lmax = max(occ.max_pattern_len, 1)
    w = max(w, 4 * Lmax)
    n, pos, out = len(hay), 0, []
    while pos < n:
        b = min(n, pos + w)
        last = b == n
        win = stream[(stream["end"] > pos) & (stream["end"] <= b)]
        sel = select(win, mk, pos, occ.max_pattern_len) if len(win) else []
        k = len(sel)
        if mk != orc.STANDARD and not last:
            while k > 0 and sel[k - 1][1] + Lmax > b:
                k -= 1
        out += sel[:k]
        floor_next = max(b + 1 - Lmax, 0)
        pos = n if last else max(sel[k - 1][2] if k else pos, floor_next)
    return out


@settings(max_examples=400, deadline=None)
@given(case(), st.sampled_from([orc.STANDARD, orc.LEFTMOST_FIRST, orc.LEFTMOST_LONGEST]), st.booleans(), st.integers(1, 40))
def test_window_rule_equals_oracle_find_iter(c, mk, casei, w):
    pats, hay = c
    if not pats or any(len(p) == 0 for p in pats):
        return
    want = orc.Oracle(pats, match_kind=mk, kind=orc.KIND_DFA, ascii_case_insensitive=casei).find_iter(hay)
    assert windowed(pats, hay, mk, casei, w) == want, (pats, hay, mk, casei, w)


def test_window_rule_on_long_inputs():
    pats2 = [b"ab", b"abab", b"b", b"ba", b"abab", b"bab", b"a", b"abababababab"]
    hay2 = np.frombuffer(b"abababbbaabab.." * 300, dtype=np.uint8).copy()
    for mk in (orc.STANDARD, orc.LEFTMOST_FIRST, orc.LEFTMOST_LONGEST):
        want = orc.Oracle(pats2, match_kind=mk, kind=orc.KIND_DFA).find_iter(hay2)
        for w in (1, 50, 97, 1000):
            assert windowed(pats2, hay2, mk, False, w) == want, (mk, w)
