"""-m gpu: the event form of the LDS walk (device/lds_emit.hip: k_lw_count_ev notes every dword that gained a record, k_lw_ev_emit
writes the ordered records without a second walk), the routing of small automata to it (host/engine_plan.hpp: lw_full), the
overlapping search that serves find_iter for pattern sets whose occurrences cannot overlap (LwHostTables::disjoint) and the
parallel block entries of the selection (device/select.hip: k_sel_entries) -- through the public calls, against the oracle."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
from aho_corasick_amd import _lib
from gpu_util import assert_same, build_pair
from oracle import orc

pytestmark = pytest.mark.gpu


def prose(n):
    import corpora
    return corpora.haystack("sherlock.txt", n)


def cases():
    rng = np.random.default_rng(11)
    yield "one byte, 1 in 26", [b"q"], rng.integers(97, 123, size=3 << 20, dtype=np.uint8)
    yield "three one-byte patterns", [b"e", b"\n", b"Z"], prose(2 << 20)
    yield "trigrams in prose", [b"the", b"you", b"and", b"ing", b"her"], prose(3 << 20)
    yield "suffixes share an end", [b"abc", b"bc", b"c", b"xbc"], rng.choice(np.frombuffer(b"abcx ", dtype=np.uint8), size=1 << 20)
    yield "nested + duplicates", [b"a", b"aa", b"aaa", b"a", b"ba", b"ab"], rng.choice(np.frombuffer(b"ab\n", dtype=np.uint8), size=1 << 19)
    yield "names", [b"Sherlock", b"Holmes", b"Watson", b"Irene", b"Adler", b"John", b"Baker"], prose(4 << 20)
    yield "sparse", [b"Moriarty", b"Lestrade"], prose(4 << 20)
    yield "no match", [b"\x00\x01", b"\x02"], prose(1 << 20)


@pytest.mark.parametrize("variants", [{}, {"lw_events": 0}, {"lw_cls": 0}, {"lw_cls": 1}, {"lw_first": 0}])
def test_overlapping_records_from_events(variants):
    for name, pats, hay in cases():
        # (a forced class form that does not fit the set leaves the automaton without the LDS walk: the other engines serve)
        a, o = build_pair(pats, "standard", {"kind": "dfa"}, variants=variants)
        dev = torch.from_numpy(hay).cuda()
        prof = _lib.CProfile()
        n0, _ = a.overlapping_device(dev, out=None, profile=prof)
        want = o.find_overlapping_iter(hay, as_numpy=True)
        ctx = f"{name} / {variants}"
        assert n0 == len(want), ctx
        if not variants or "lw_events" in variants:
            assert prof.engine_used == 3, ctx   # the LDS walk first (host/engine_plan.hpp)
        assert_same(a.find_overlapping_iter(dev, as_numpy=True), want, ctx + " / device haystack, host records")
        assert_same(a.find_overlapping_iter(hay, as_numpy=True), want, ctx + " / host haystack")
        out = torch.empty(len(want) * 24 + 4096, dtype=torch.uint8, device="cuda")
        n, ok = a.overlapping_device(dev, out=out)
        assert ok and n == len(want), ctx
        assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, ctx + " / device records")
        # a buffer one record short: the count comes back, nothing usable is written
        if len(want) > 1:
            small = torch.empty((len(want) - 1) * 24, dtype=torch.uint8, device="cuda")
            n, ok = a.overlapping_device(dev, out=small)
            assert not ok and n == len(want), ctx
        # spans and shards that start and end off every grid, on a misaligned device pointer
        for off, lo, hi in ((1, 0, len(hay) - 1), (37, 1000, len(hay) - 777), (63, 511, 513), (5, 4096, 4096 + 1537)):
            h2 = hay[off:]
            d2 = dev[off:]
            hi2 = min(hi, len(h2))
            sub = o.find_overlapping_iter(h2, span=(lo, hi2), as_numpy=True)
            n, ok = a.overlapping_device(d2, span=(lo, hi2), out=out)
            assert ok and n == len(sub), f"{ctx} / span {off} {lo} {hi2}"
            assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), sub, f"{ctx} / span {off} {lo} {hi2}")
        mid = len(hay) // 2 + 13
        parts = []
        for sb, se in ((0, mid), (mid, len(hay))):
            n, ok = a.overlapping_device(dev, shard=(sb, se), out=out)
            assert ok
            parts.append(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE).copy())
        assert_same(np.concatenate(parts), want, ctx + " / two shards")


def test_tiny_haystacks_are_all_edge():
    pats = [b"ab", b"b", b"abc"]
    a, o = build_pair(pats, "standard", {"kind": "dfa"})
    rng = np.random.default_rng(3)
    for n in list(range(0, 40)) + [63, 64, 65, 511, 512, 513, 1023, 1025, 32767, 32768, 32769, 65537]:
        hay = rng.choice(np.frombuffer(b"abc", dtype=np.uint8), size=n).astype(np.uint8)
        want = o.find_overlapping_iter(hay, as_numpy=True)
        assert_same(a.find_overlapping_iter(torch.from_numpy(hay).cuda() if n else hay, as_numpy=True), want, f"len {n}")


def test_event_list_overflow_falls_back_to_the_chunk_fill():
    """a record at every byte: the events (one per dword) exceed the slabs (one per 16 bytes) and k_lw_fill serves"""
    hay = np.frombuffer(b"ab" * (1 << 20), dtype=np.uint8).copy()
    a, o = build_pair([b"a", b"b", b"ab"], "standard", {"kind": "dfa"})
    want = o.find_overlapping_iter(hay, as_numpy=True)
    dev = torch.from_numpy(hay).cuda()
    out = torch.empty(len(want) * 24, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        n, ok = a.overlapping_device(dev, out=out)
        assert ok and n == len(want)
        assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, "overflow, device records")
    assert_same(a.find_overlapping_iter(dev, as_numpy=True), want, "overflow, host records")


def test_long_patterns_take_a_larger_lane_chunk():
    rng = np.random.default_rng(2)
    pats = [bytes(rng.integers(97, 101, size=n, dtype=np.uint8)) for n in (90, 150, 33)]
    hay = rng.integers(97, 101, size=2 << 20, dtype=np.uint8)
    for k, pos in enumerate(range(1000, len(hay) - 200, 40961)):
        p = np.frombuffer(pats[k % 3], dtype=np.uint8)
        hay[pos:pos + len(p)] = p
    a, o = build_pair(pats, "standard", {"kind": "dfa"})
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > 40
    assert_same(a.find_overlapping_iter(torch.from_numpy(hay).cuda(), as_numpy=True), want, "long patterns")


@pytest.mark.parametrize("mk", ["standard", "leftmost_first", "leftmost_longest"])
def test_find_iter_of_disjoint_sets_is_the_overlapping_search(mk):
    rng = np.random.default_rng(4)
    for name, pats, hay in (("one byte", [b"a"], rng.integers(97, 123, size=1 << 20, dtype=np.uint8)),
                            ("letters", [b"e", b"t", b"a", b"o", b"\n"], prose(2 << 20)),
                            ("isolated words", [b"xq", b"zj"], rng.choice(np.frombuffer(b"xqzj ab", dtype=np.uint8), size=1 << 20))):
        for variants in ({}, {"find_iter_disjoint": 0}):
            a, o = build_pair(pats, mk, variants=variants)
            want = o.find_iter(hay, as_numpy=True)
            dev = torch.from_numpy(hay).cuda()
            assert_same(a.find_iter(dev, as_numpy=True), want, f"{name} / {mk} / {variants}")
            out = torch.empty(len(want) * 24 + 240, dtype=torch.uint8, device="cuda")
            n, ok = a.find_iter_device(dev, out)
            assert ok and n == len(want)
            assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, f"{name} / {mk} / {variants} / device records")
            sub = o.find_iter(hay, span=(777, len(hay) - 5), as_numpy=True)
            assert_same(a.find_iter(ac.Input(dev).range(777, len(hay) - 5), as_numpy=True), sub, f"{name} / {mk} / span")


@pytest.mark.parametrize("mk", ["standard", "leftmost_first", "leftmost_longest"])
def test_selection_entries_from_breaks_and_without_them(mk):
    """k_sel_entries: natural text has a break every few occurrences; a periodic text under a self-overlapping pattern has
    none in a whole block of 1 024 occurrences, and the serial hop takes over"""
    rng = np.random.default_rng(6)
    periodic = np.frombuffer(b"ab" * (1 << 18), dtype=np.uint8).copy()
    mixed = np.concatenate([prose(1 << 20), periodic[: 1 << 18], prose(1 << 19), np.frombuffer(b"z" * 70000, dtype=np.uint8)])
    for name, pats, hay in (("trigrams", [b"the", b"he ", b"and", b"nd ", b"ing"], prose(3 << 20)),
                            ("self-overlapping, periodic text", [b"abab", b"bab"], periodic),
                            ("mixed", [b"abab", b"the", b"zzzzzzzzzz", b"e t"], mixed)):
        a, o = build_pair(pats, mk, variants={"start_table": 0})   # (the occurrence stream + selection, however dense)
        want = o.find_iter(hay, as_numpy=True)
        assert len(want) > 1024, name
        dev = torch.from_numpy(hay).cuda()
        for _ in range(2):
            assert_same(a.find_iter(dev, as_numpy=True), want, f"{name} / {mk}")


def test_enqueue_only_form_of_the_event_walk():
    """acgpu_find_overlapping_enqueue on a small automaton: count walk + events -> one-workgroup scan -> emit, nothing decided on
    the host.  Sparse results are delivered; a result whose events overflow their slabs is REPORTED (totals[1] beyond
    ENQUEUE_MAX_EVENTS: repeat synchronously) unless the chunk fill is queued behind the events -- after a synchronous call
    met the overflow, with classic=True, or without adaptive hints."""
    import corpora
    text = corpora.haystack("sherlock.txt", 4 << 20)
    dense = np.frombuffer(b"ab" * (1 << 20), dtype=np.uint8).copy()
    pats = [b"a", b"b", b"the", b"Holmes"]
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    want_text, want_dense = o.find_overlapping_iter(text, as_numpy=True), o.find_overlapping_iter(dense, as_numpy=True)
    d_text, d_dense = torch.from_numpy(text).cuda(), torch.from_numpy(dense).cuda()
    out = torch.empty(max(len(want_text), len(want_dense)) * 24 + 240, dtype=torch.uint8, device="cuda")
    tot = torch.zeros(2, dtype=torch.int64, device="cuda")

    def enq(a, d, **kw):
        a.overlapping_enqueue(d, out, tot, **kw)
        torch.cuda.synchronize()
        t = tot.cpu().numpy().view(np.uint64)
        return int(t[0]), int(t[1])

    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build(pats)
    for _ in range(2):
        n, ev = enq(a, d_text)
        assert (n, ev) == (len(want_text), 0)
        assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), want_text, "enqueue, prose")
    n, ev = enq(a, d_dense)                      # overflow, nobody fills: reported
    assert n == len(want_dense) and ev > a.ENQUEUE_MAX_EVENTS
    n, ev = enq(a, d_dense, classic=True)        # dense results asked for: the chunk fill is queued
    assert (n, ev) == (len(want_dense), 0)
    assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), want_dense, "enqueue classic, dense")
    m, ok = a.overlapping_device(d_dense, out=out)   # the synchronous call fills, and remembers
    assert ok and m == len(want_dense)
    n, ev = enq(a, d_dense)
    assert (n, ev) == (len(want_dense), 0)
    assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), want_dense, "enqueue after a synchronous overflow")
    n, ev = enq(a, d_text, span=(1234, len(text) - 77))
    sub = o.find_overlapping_iter(text, span=(1234, len(text) - 77), as_numpy=True)
    assert (n, ev) == (len(sub), 0)
    assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), sub, "enqueue, span")
    det = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_deterministic_routing(True).build(pats)
    n, ev = enq(det, d_dense)                    # no hints: the fill is always queued
    assert (n, ev) == (len(want_dense), 0)
    assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), want_dense, "enqueue, deterministic routing")
