"""Product host builder (aho-corasick_amd/csrc/host/builder.cpp) vs the oracle: identical state
numbering, failure links, match-list order, DFA words and contiguous-NFA words."""
import ctypes as C
import random

import numpy as np
import pytest

import aho_corasick_amd as ac
import refmatrix
from oracle import orc

MK = {"standard": 0, "leftmost_first": 1, "leftmost_longest": 2}
KIND = {None: None, "nnfa": ac.AhoCorasickKind.NoncontiguousNFA, "cnfa": ac.AhoCorasickKind.ContiguousNFA,
        "dfa": ac.AhoCorasickKind.DFA}
OKIND = {None: orc.KIND_AUTO, "nnfa": orc.KIND_NNFA, "cnfa": orc.KIND_CNFA, "dfa": orc.KIND_DFA}
SK = {"both": 0, "unanchored": 1, "anchored": 2}


def arr(ptr, n):
    if n == 0:
        return np.zeros(0, dtype=np.uint32)
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


def compare(pats, mk=0, kind=None, start="unanchored", casei=False, byte_classes=True, dense_depth=None):
    b = ac.AhoCorasick.builder().match_kind(mk).start_kind(SK[start]).kind(KIND[kind]) \
        .ascii_case_insensitive(casei).byte_classes(byte_classes)
    if dense_depth is not None:
        b.dense_depth(dense_depth)
    a = b.build(pats)
    o = orc.Oracle(pats, match_kind=mk, start_kind=SK[start], kind=OKIND[kind], ascii_case_insensitive=casei,
                   byte_classes=byte_classes, dense_depth=dense_depth)
    assert int(a.kind()) == o.kind
    assert a.patterns_len() == o.patterns_len
    assert a.min_pattern_len() == o.min_pattern_len and a.max_pattern_len() == o.max_pattern_len
    assert a.memory_usage() == o.memory_usage
    ta, to = a.tables(), o.tables()
    assert ta.nnfa_states == to.nnfa_states
    assert (ta.nnfa_max_match_id, ta.nnfa_start_unanchored_id, ta.nnfa_start_anchored_id) == \
        (to.max_match_id, to.start_unanchored_id, to.start_anchored_id)
    assert bytes(ta.byte_classes) == bytes(to.byte_classes)
    n = ta.nnfa_states
    fail, depth = arr(ta.nnfa_fail, n), arr(ta.nnfa_depth, n)
    moff = arr(ta.nnfa_match_off, n + 1)
    mpid = arr(ta.nnfa_match_pid, int(moff[-1]))
    for s in ([0, 1, 2, 3] + random.sample(range(n), min(n, 300)) if n > 400 else range(n)):
        of, od, ol = o.nnfa_state(s)
        assert (int(fail[s]), int(depth[s])) == (of, od), s
        assert list(mpid[moff[s]:moff[s + 1]]) == ol, s
    assert np.array_equal(arr(ta.pattern_lens, a.patterns_len()), arr(to.pattern_lens, o.patterns_len))
    if o.kind == orc.KIND_DFA:
        assert (ta.dfa_state_len, ta.dfa_stride2, ta.dfa_trans_len) == (to.dfa_state_len, to.dfa_stride2, to.dfa_trans_len)
        assert (ta.dfa_max_match_id, ta.dfa_start_unanchored_id, ta.dfa_start_anchored_id) == \
            (to.dfa_max_match_id, to.dfa_start_unanchored_id, to.dfa_start_anchored_id)
        assert np.array_equal(arr(ta.dfa_trans, ta.dfa_trans_len), arr(to.dfa_trans, to.dfa_trans_len))
        nm = ta.dfa_num_match_states
        assert nm == to.dfa_num_match_states
        offa, offo = arr(ta.dfa_match_off, nm + 1), arr(to.dfa_match_off, nm + 1)
        assert np.array_equal(offa, offo)
        assert np.array_equal(arr(ta.dfa_match_pid, int(offa[-1])), arr(to.dfa_match_pid, int(offo[-1])))
    if o.kind == orc.KIND_CNFA:
        assert ta.cnfa_repr_len == to.cnfa_repr_len
        assert np.array_equal(arr(ta.cnfa_repr, ta.cnfa_repr_len), arr(to.cnfa_repr, to.cnfa_repr_len))
        assert (ta.cnfa_max_match_id, ta.cnfa_start_unanchored_id, ta.cnfa_start_anchored_id) == \
            (to.cnfa_max_match_id, to.cnfa_start_unanchored_id, to.cnfa_start_anchored_id)


CASES = list(refmatrix.all_cases())


@pytest.mark.parametrize("cid,mk,api,kw,vectors", CASES, ids=[c[0] for c in CASES])
def test_tables_on_reference_pattern_sets(cid, mk, api, kw, vectors):
    seen = set()
    for v in vectors:
        pats, _, _ = refmatrix.unhex(v)
        key = tuple(pats)
        if key in seen:
            continue
        seen.add(key)
        compare(pats, MK[mk], kw.get("kind"), kw.get("start_kind", "unanchored"),
                kw.get("ascii_case_insensitive", False), kw.get("byte_classes", True), kw.get("dense_depth"))


@pytest.mark.parametrize("kind", ["dfa", "cnfa", "nnfa"])
def test_tables_c2_pattern_set(kind):
    pats = orc.gen_patterns(1000)
    compare(pats, 0, kind)
    compare(pats, 1, kind, casei=True)          # C5: LeftmostFirst + ascii_case_insensitive
    if kind == "dfa":
        compare(pats, 0, kind, byte_classes=False)  # the "256-wide" table
        compare(pats, 0, kind, start="both")


def test_tables_c4_pattern_set():
    compare(orc.gen_patterns(100000, seed=0xAC04), 0, "cnfa")


def test_tables_random_small_alphabet():
    rng = random.Random(12345)
    for it in range(300):
        alpha = rng.choice([b"ab", b"abc", b"aAbB", b"\x00\x01\xff", b"abcdefgh"])
        npat = rng.randint(0, 12)
        pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(0, 6))) for _ in range(npat)]
        mk = rng.randint(0, 2)
        kind = rng.choice(["dfa", "cnfa", "nnfa", None])
        start = rng.choice(["unanchored", "anchored", "both"])
        compare(pats, mk, kind, start, casei=rng.random() < 0.3, byte_classes=rng.random() < 0.7,
                dense_depth=rng.choice([None, 0, 1, 2, 5, 0xFFFFFFFF]))
