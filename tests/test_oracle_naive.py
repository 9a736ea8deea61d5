"""Second opinion on the oracle: a naive matcher (reference DESIGN.md:60-63 semantics) under hypothesis.

Overlapping (Standard), non-empty patterns: every occurrence, ordered by (end, -len, pid).
LeftmostFirst / LeftmostLongest find_iter: smallest start >= pos, then lowest pid / greatest length.
Standard find_iter: among occurrences with start >= pos the smallest end, first in stream order.
Chunked scanning with an L-1 halo reproduces the unchunked stream (the seam rule of SURVEY.md 8e)."""
from hypothesis import given, settings, strategies as st

from oracle import orc

ALPHA = st.sampled_from([b"ab", b"abc", b"aAbB"])


def occurrences(pats, hay, casei=False):
    h = hay.lower() if casei else hay
    out = []
    for pid, p in enumerate(pats):
        q = p.lower() if casei else p
        for s in range(len(h) - len(q) + 1):
            if h[s:s + len(q)] == q:
                out.append((pid, s, s + len(q)))
    return out


@st.composite
def case(draw, allow_empty=False):
    alpha = draw(ALPHA)
    byte = st.sampled_from(list(alpha))
    pats = draw(st.lists(st.binary(min_size=0 if allow_empty else 1, max_size=5).map(
        lambda b: bytes(alpha[x % len(alpha)] for x in b)), min_size=0, max_size=8))
    hay = bytes(draw(st.lists(byte, min_size=0, max_size=60)))
    return pats, hay


@settings(max_examples=300, deadline=None)
@given(case(), st.sampled_from([orc.KIND_NNFA, orc.KIND_CNFA, orc.KIND_DFA]), st.booleans())
def test_overlapping_is_all_occurrences_in_order(c, kind, casei):
    pats, hay = c
    want = sorted(occurrences(pats, hay, casei), key=lambda m: (m[2], -(m[2] - m[1]), m[0]))
    got = orc.Oracle(pats, kind=kind, ascii_case_insensitive=casei).find_overlapping_iter(hay)
    assert got == want


@settings(max_examples=300, deadline=None)
@given(case(), st.sampled_from([orc.KIND_NNFA, orc.KIND_CNFA, orc.KIND_DFA]),
       st.sampled_from([orc.STANDARD, orc.LEFTMOST_FIRST, orc.LEFTMOST_LONGEST]))
def test_find_iter_semantics(c, kind, mk):
    pats, hay = c
    occ = occurrences(pats, hay)
    stream = sorted(occ, key=lambda m: (m[2], -(m[2] - m[1]), m[0]))
    want, pos = [], 0
    while True:
        cand = [m for m in stream if m[1] >= pos]
        if not cand:
            break
        if mk == orc.STANDARD:
            m = min(cand, key=lambda m: m[2])  # min is stable: first in stream order among equal ends
        elif mk == orc.LEFTMOST_FIRST:
            m = min(cand, key=lambda m: (m[1], m[0]))
        else:
            m = min(cand, key=lambda m: (m[1], -(m[2] - m[1]), m[0]))
        want.append(m)
        pos = m[2]
    got = orc.Oracle(pats, kind=kind, match_kind=mk).find_iter(hay)
    assert got == want


@settings(max_examples=200, deadline=None)
@given(case(allow_empty=True), st.integers(1, 9))
def test_seam_rule_reproduces_unchunked_stream(c, chunk):
    pats, hay = c
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    full = o.find_overlapping_iter(hay)
    halo = max(o.max_pattern_len, 1) - 1 if pats else 0
    got = []
    for lo in range(0, max(len(hay), 1), chunk):
        hi = min(lo + chunk, len(hay))
        w = max(0, lo - halo)
        part = o.find_overlapping_iter(hay, span=(w, hi))
        if lo == 0:
            got += part                                   # first chunk owns the start-state matches too
        else:
            got += [m for m in part if m[2] > lo]          # end in (lo, hi]
    assert got == full


@settings(max_examples=400, deadline=None)
@given(case(), st.sampled_from([1, 2]), st.booleans())
def test_earliest_on_a_leftmost_automaton_is_the_standard_iteration(c, mk, casei):
    """Input::earliest makes every search return at the first match state it enters (src/automaton.rs:1266).  Up to that
    state a leftmost automaton is the Standard one over the same patterns (noncontiguous.rs:1296-1346 changes failure links
    behind match states only), so find / find_iter with `earliest` on a leftmost automaton equal the Standard automaton's --
    the rule the device uses to serve them from the occurrence stream (capi_find.cpp)."""
    pats, hay = c
    std = orc.Oracle(pats, match_kind=0, ascii_case_insensitive=casei)
    lm = orc.Oracle(pats, match_kind=mk, ascii_case_insensitive=casei)
    assert lm.find_iter(hay, earliest=True) == std.find_iter(hay)
    assert lm.find(hay, earliest=True) == std.find(hay)
