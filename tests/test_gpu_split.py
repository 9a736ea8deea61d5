"""-m gpu: split pattern sets (capi_overlap.cpp: overlapping_split) -- a dictionary of long words plus a few short stragglers is
searched as two automata whose ordered record streams are merged on the device (device/merge.hip).  Every public search
that rests on the overlapping stream -- overlapping (device / host haystack, device / host output, spans, shards, the
enqueue form's hand-back), find_iter under the three match kinds, replace_all, the stream search -- against the oracle."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
import corpora
from gpu_util import MK, assert_same, build_pair
from oracle import orc

pytestmark = pytest.mark.gpu

STRAGGLERS = [b"cab", b"the", b"lamp", b"e", b"Holmes", b"window", b"I", b"that"]


def patterns(k):
    words = list(corpora.words("words-5000"))
    # nested with dictionary words and with each other, a duplicate of a straggler and of a long word
    return words + STRAGGLERS[:k] + ([STRAGGLERS[0], words[17]] if k else [])


def text(n):
    t = corpora.haystack("sherlock.txt")
    return np.tile(t, -(-n // len(t)))[:n].copy()


@pytest.mark.parametrize("k", [1, 4, 8])
def test_overlapping_records(k):
    pats = patterns(k)
    hay = text(6 << 20)
    a, o = build_pair(pats, "standard", {"kind": None})
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > 5_000
    dev = torch.from_numpy(hay).cuda()
    assert_same(a.find_overlapping_iter(dev, as_numpy=True), want, f"split k={k} device haystack")
    assert_same(a.find_overlapping_iter(hay, as_numpy=True), want, f"split k={k} host haystack")
    out = torch.empty(len(want) * 24 + 4096, dtype=torch.uint8, device="cuda")
    n, ok = a.overlapping_device(dev, out=out)
    assert ok and n == len(want)
    assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, "device output")
    n, ok = a.overlapping_device(dev, out=out[: 24 * 100])   # too small: the count is still reported
    assert not ok and n == len(want)
    # counts without a buffer, and buffers that hold one part's stream but not both: nothing beyond the caller's room is
    # materialised (round 5 built both streams first), the count comes back all the same
    n, ok = a.overlapping_device(dev, out=None)
    assert n == len(want) and not ok
    long_only = sum(1 for p in want["pattern"] if p < 5000)
    for room in (1, long_only, long_only + 1, len(want) - 1):
        n, ok = a.overlapping_device(dev, out=out[: 24 * room])
        assert not ok and n == len(want), room
    n, ok = a.overlapping_device(dev, out=out[: 24 * len(want)])
    assert ok and n == len(want)
    assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, "device output, exact room")
    # a span, and two shards of it whose concatenation is the span's stream
    lo, hi = 12345, len(hay) - 4321
    sub = o.find_overlapping_iter(hay, span=(lo, hi), as_numpy=True)
    mid = (lo + hi) // 2 + 3
    parts = []
    for sb, se in ((lo, mid), (mid, hi)):
        n, ok = a.overlapping_device(dev, span=(lo, hi), shard=(sb, se), out=out)
        assert ok
        parts.append(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE).copy())
    assert_same(np.concatenate(parts), sub, "two shards of a span")
    # the enqueue-only form hands a split set back to the synchronous call
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    a.overlapping_enqueue(dev, out, totals)
    torch.cuda.synchronize()
    assert int(totals.cpu().numpy().view(np.uint64)[1]) > a.ENQUEUE_MAX_EVENTS


@pytest.mark.parametrize("mk", ["standard", "leftmost_first", "leftmost_longest"])
def test_find_iter_and_replace_all(mk):
    pats = patterns(4)
    hay = text(3 << 20)
    a, o = build_pair(pats, mk, {"kind": None})
    want = o.find_iter(hay, as_numpy=True)
    dev = torch.from_numpy(hay).cuda()
    assert_same(a.find_iter(dev, as_numpy=True), want, f"split find_iter {mk}")
    repl = [b"<%d>" % (i % 11) for i in range(len(pats))]
    small = hay[: 1 << 20]
    got = bytes(a.replace_all_bytes(torch.from_numpy(small).cuda(), repl).cpu().numpy())
    assert got == orc.replace_all_bytes(o, small, repl)


def test_stream_search():
    import io
    from gpu_util import triples
    pats = patterns(2)
    hay = text(2 << 20)
    a, o = build_pair(pats, "standard", {"kind": None})
    want = triples(o.find_iter(hay, as_numpy=True))
    got = [(m.pattern(), m.start(), m.end()) for m in a.stream_find_iter(io.BytesIO(hay.tobytes()), chunk_bytes=300_001)]
    assert got == want and len(want) > 1000


def test_sets_that_are_not_split_still_agree():
    """65 stragglers (one too many), or a shortest pattern of seven bytes: the unsplit engines."""
    words = list(corpora.words("words-5000"))
    hay = text(2 << 20)
    for pats in (words + [b"w%02d" % i for i in range(65)], words + [b"morning"]):
        a, o = build_pair(pats, "standard", {"kind": None})
        assert_same(a.find_overlapping_iter(torch.from_numpy(hay).cuda(), as_numpy=True), o.find_overlapping_iter(hay, as_numpy=True), "unsplit")
