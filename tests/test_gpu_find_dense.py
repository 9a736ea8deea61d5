"""-m gpu: leftmost find_iter through the per-start candidate table (device/start_select.hip) vs the oracle's FindIter
(src/automaton.rs:857-936): occurrence-dense inputs (an occurrence per byte and more), chains that never merge ('aa' over
a run of 'a'), windows chained through their exit offsets, blocks crossed by long matches, both leftmost kinds,
case-insensitive sets, sub-spans, host and device output, and the automatic switch from the occurrence-stream form."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
from aho_corasick_amd import _lib
from gpu_util import assert_same, build_pair
from oracle import orc

pytestmark = pytest.mark.gpu
LEFTMOST = ["leftmost_first", "leftmost_longest"]


def dev(h):
    return torch.from_numpy(np.ascontiguousarray(h)).cuda() if len(h) else torch.zeros(0, dtype=torch.uint8, device="cuda")


def check(a, o, hay, span=None, ctx=""):
    dh = dev(hay)
    want = o.find_iter(hay, span=span, as_numpy=True)
    got = a.find_iter(ac.Input(dh).range(*span) if span else dh, as_numpy=True)
    assert_same(got, want, ctx + " (host output)")
    out = torch.empty(max(len(want), 1) * 24, dtype=torch.uint8, device="cuda")
    n, ok = a.find_iter_device(dh, out, span=span)
    assert ok and n == len(want), (ctx, n, len(want))
    assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, ctx + " (device output)")
    return len(want)


@pytest.fixture
def table():
    """engine variants of the automata under test: leftmost find_iter from the per-start table whatever the density"""
    return {"find_iter_start_table": 1}


@pytest.mark.parametrize("mk", LEFTMOST)
def test_runs_of_one_byte(mk, table):
    """an occurrence per byte; 'aa' over a run: two disjoint chains (even / odd starts) that never merge"""
    n = 300_001
    hay = np.full(n, 0x61, dtype=np.uint8)
    hay[100_000:100_007] = 0x62
    for pats in ([b"a"], [b"aa"], [b"a", b"aa"], [b"aa", b"a"], [b"aaa", b"aa", b"ab"], [b"b", b"ba", b"aaaaaaa"]):
        a, o = build_pair(pats, mk, variants=table)
        assert check(a, o, hay, ctx=f"{mk} {pats}") > 40_000
        check(a, o, hay, span=(1, n - 1), ctx=f"{mk} {pats} span")
        check(a, o, hay, span=(99_990, 100_020), ctx=f"{mk} {pats} short span")


@pytest.mark.parametrize("mk", LEFTMOST)
@pytest.mark.parametrize("window_kib", [1, 4, 64])
def test_windows_are_chained(mk, window_kib, table):
    """windows of 1 024 / 4 096 / 65 536 positions: the chain enters each at the exit offset of its predecessor"""
    table = dict(table, ss_window_kib=window_kib)
    rng = np.random.default_rng(window_kib)
    pats = [bytes(rng.integers(0x61, 0x64, size=int(rng.integers(1, 9)), dtype=np.uint8)) for _ in range(40)] + [b"abcabcabcabcabcabcabcabc"]
    a, o = build_pair(pats, mk, variants=table)
    hay = rng.integers(0x61, 0x65, size=200_000, dtype=np.uint8)
    hay[70_000:70_400] = np.frombuffer(b"abc" * 134, dtype=np.uint8)[:400]
    check(a, o, hay, ctx=f"{mk} window {window_kib} KiB")
    check(a, o, hay, span=(1023, 150_001), ctx=f"{mk} window {window_kib} KiB span")


@pytest.mark.parametrize("mk", LEFTMOST)
def test_random_dense_sets(mk, table):
    rng = np.random.default_rng(91 + len(mk))
    for case in range(40):
        sigma = int(rng.integers(1, 5))
        pats = [bytes(rng.integers(0x61, 0x61 + sigma, size=int(rng.integers(1, 7)), dtype=np.uint8))
                for _ in range(int(rng.integers(1, 30)))]
        a, o = build_pair(pats, mk, {"kind": [None, "dfa", "cnfa", "nnfa"][case % 4], "ascii_case_insensitive": case % 5 == 0}, variants=table)
        n = int(rng.integers(0, 6000))
        hay = rng.integers(0x61, 0x61 + sigma + int(rng.integers(0, 3)), size=n, dtype=np.uint8)
        if case % 5 == 0 and n:
            hay[rng.integers(0, n, size=n // 3)] -= 0x20     # upper-case letters
        check(a, o, hay, ctx=f"case {case}")
        if n > 10:
            s = int(rng.integers(0, n)); e = int(rng.integers(s, n + 1))
            check(a, o, hay, span=(s, e), ctx=f"case {case} span")


@pytest.mark.parametrize("mk", LEFTMOST)
def test_long_patterns_cross_blocks(mk, table):
    """matches of up to 1 024 bytes (the longest the table serves) leave a block far behind its end"""
    rng = np.random.default_rng(5)
    long1 = bytes(rng.integers(0x61, 0x63, size=1024, dtype=np.uint8))
    long2 = bytes(rng.integers(0x61, 0x63, size=700, dtype=np.uint8))
    pats = [long1, long2, long1[:300], b"ab", b"b", long2[5:90]]
    a, o = build_pair(pats, mk, variants=table)
    hay = rng.integers(0x61, 0x63, size=120_000, dtype=np.uint8)
    for at, p in ((500, long1), (1020, long2), (3000, long1), (4090, long1[:300]), (50_000, long2), (118_976, long1)):
        hay[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
    check(a, o, hay, ctx=mk)
    a, o = build_pair(pats, mk, variants=dict(table, ss_window_kib=2))
    check(a, o, hay, ctx=mk + " small windows")


def test_a_longer_pattern_is_not_served_by_the_table(table):
    pats = [b"a" * 1025, b"a", b"ab"]
    a, o = build_pair(pats, "leftmost_first", variants=table)
    hay = np.full(5000, 0x61, dtype=np.uint8)
    check(a, o, hay)


@pytest.mark.parametrize("mk", LEFTMOST)
def test_switches_by_itself_on_dense_input_and_back(mk):
    """no knob: the occurrence-stream form meets more than one occurrence per 64 bytes, the call is answered from the table,
    and so are the next ones while results stay dense; a sparse result sends the automaton back to the filters"""
    pats = [b"e", b"th", b"the", b"and", b"zebra crossing"]
    a, o = build_pair(pats, mk)
    rng = np.random.default_rng(3)
    text = np.frombuffer(b"the quick brown fox jumps over the lazy dog and then rests ", dtype=np.uint8)
    dense = np.tile(text, 40_000)
    sparse = rng.integers(0x30, 0x3A, size=len(dense), dtype=np.uint8)
    sparse[1_000_000:1_000_014] = np.frombuffer(b"zebra crossing", dtype=np.uint8)
    p = _lib.CProfile()
    for hay, expect_many in ((dense, True), (dense, True), (sparse, False), (sparse, False), (dense, True)):
        dh = dev(hay)
        want = o.find_iter(hay, as_numpy=True)
        got = a.find_iter(dh, as_numpy=True, profile=p)
        assert_same(got, want, f"{mk} many={expect_many}")
        assert (len(want) > 100_000) == expect_many


def test_output_buffer_too_small_reports_the_count(table):
    a, o = build_pair([b"a", b"b"], "leftmost_first", variants=table)
    hay = np.frombuffer(b"ab" * 5000, dtype=np.uint8)
    out = torch.empty(100 * 24, dtype=torch.uint8, device="cuda")
    n, ok = a.find_iter_device(dev(hay), out)
    assert not ok and n == 10_000
