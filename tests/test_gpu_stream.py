"""-m gpu: stream search (acgpu_stream_*) vs the oracle's Standard find_iter of the whole stream
(StreamChunkIter, src/automaton.rs:1036-1244, reports exactly that sequence)."""
import io

import numpy as np
import pytest

import aho_corasick_amd as ac
from gpu_util import assert_same, build_pair
from oracle import orc

pytestmark = pytest.mark.gpu


def triples(ms):
    return [(m.pattern(), m.start(), m.end()) for m in ms]


def want_triples(o, hay):
    a = o.find_iter(hay, as_numpy=True)
    return list(zip(a["pattern"].tolist(), a["start"].tolist(), a["end"].tolist()))


def test_reference_doc_examples():
    a = ac.AhoCorasick.new(["append", "appendage", "app"])                    # src/ahocorasick.rs:884-900
    got = triples(a.stream_find_iter(io.BytesIO(b"append the app to the appendage")))
    assert [t[0] for t in got] == [2, 2, 2]                                   # the documented pattern ids
    assert got == [(2, 0, 3), (2, 11, 14), (2, 22, 25)]                       # + their spans
    a = ac.AhoCorasick.new(["fox", "brown", "quick"])                         # README.md:85-99
    w = io.BytesIO()
    a.stream_replace_all(io.BytesIO(b"The quick brown fox."), w, ["sloth", "grey", "slow"])
    assert w.getvalue() == b"The slow grey sloth."


def test_stream_errors():
    with pytest.raises(ac.MatchError):                                        # src/automaton.rs:1067-1069
        list(ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).build(["a"]).stream_find_iter(io.BytesIO(b"a")))
    with pytest.raises(ac.MatchError):                                        # :1082-1084
        list(ac.AhoCorasick.new(["", "a"]).stream_find_iter(io.BytesIO(b"a")))


@pytest.mark.parametrize("chunk", [1, 7, 64, 1000, 4096, 1 << 20])
def test_stream_equals_find_iter(chunk):
    n = 200_000 if chunk >= 64 else 3000
    hay = orc.gen_haystack(0, n, seed=0xAC08, lo=0x61, span=26)
    pats = [p[:k] for p, k in zip(orc.gen_patterns(200, seed=3, lo=0x61, span=26), [2, 3, 4, 5, 9] * 40)]
    a, o = build_pair(pats, "standard")
    want = want_triples(o, hay)
    assert len(want) > 100
    assert triples(a.stream_find_iter(io.BytesIO(hay.tobytes()), chunk_bytes=chunk)) == want
    repl = [b"<%d>" % i for i in range(len(pats))]
    w = io.BytesIO()
    a.stream_replace_all(io.BytesIO(hay.tobytes()), w, repl, chunk_bytes=chunk)
    assert w.getvalue() == orc.replace_all_bytes(o, hay, repl)


def test_stream_randomized_small():
    rng = np.random.default_rng(21)
    for case in range(30):
        sigma = int(rng.integers(2, 5))
        pats = [bytes(rng.integers(0x61, 0x61 + sigma, size=int(rng.integers(1, 7)), dtype=np.uint8))
                for _ in range(int(rng.integers(1, 10)))]
        a, o = build_pair(pats, "standard", {"kind": [None, "dfa", "cnfa"][case % 3]})
        hay = rng.integers(0x61, 0x61 + sigma + 1, size=int(rng.integers(0, 2000)), dtype=np.uint8)
        chunk = int(rng.integers(1, 300))
        assert triples(a.stream_find_iter(io.BytesIO(hay.tobytes()), chunk_bytes=chunk)) == want_triples(o, hay), \
            f"case {case} pats={pats} chunk={chunk}"


def test_stream_feed_split_when_the_chunk_does_not_fit():
    """A fed chunk whose occurrence stream exhausts device memory is fed as two halves (recursively): forced here for
    every feed above 64 KiB.  Dense input on purpose (the guard that sends find_iter to the serial loop is off inside
    the stream search, which has no serial form)."""
    pats = [b"a"] * 40 + [b"b"] * 40 + [b"ab", b"abba", b"bbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbb"]
    a, o = build_pair(pats, "standard", variants={"stream_split": 1})
    hay = np.random.default_rng(5).integers(0x61, 0x63, size=700_001, dtype=np.uint8)
    hay[300_000:300_040] = 0x62
    want = want_triples(o, hay)
    got = triples(a.stream_find_iter(io.BytesIO(hay.tobytes()), chunk_bytes=1 << 19))
    assert got == want
    a, o = build_pair(pats, "standard")
    assert triples(a.stream_find_iter(io.BytesIO(hay.tobytes()), chunk_bytes=1 << 19)) == want   # unsplit: same


def test_host_haystack_and_feed_pipelined_under_the_copy(monkeypatch):
    """Large HOST inputs are searched piece by piece while a helper thread copies the later pieces (HostPipe,
    capi.cpp): forced here with 1 MiB pieces -- overlapping search (whole span, sub-span, shard, too-small buffer), and
    a stream fed with one large host chunk after a small one."""
    monkeypatch.setenv("ACGPU_HOST_PIECE_MIB", "1")
    pats = orc.gen_patterns(1000, seed=0xAC01)
    n = (9 << 20) + 12345
    hay = orc.gen_haystack(0, n, seed=0xAC02)
    from gpu_util import plant
    plant(hay, pats[:64], [(k << 20) - d for k in range(1, 10) for d in (0, 1, 7, 15)] + [3000 * k for k in range(1, 200)])
    a, o = build_pair(pats, "standard", {"kind": "dfa"})
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > 200
    assert_same(a.find_overlapping_iter(hay, as_numpy=True), want, "host, pipelined")
    for span in [(5, n - 3), ((1 << 20) - 4, (6 << 20) + 9)]:
        assert_same(a.find_overlapping_iter(ac.Input(hay).range(*span), as_numpy=True),
                    o.find_overlapping_iter(hay, span=span, as_numpy=True), f"host span {span}")
    mid = (4 << 20) + 77
    parts = [a.find_overlapping_shard(ac.Input(hay), 0, mid), a.find_overlapping_shard(ac.Input(hay), mid, n)]
    assert_same(np.concatenate(parts), want, "host shards, pipelined")
    # dense result: BUFFER_TOO_SMALL then the retry with the reported size (the binding does that)
    az = orc.gen_patterns(300, seed=5, lo=0x61, span=26)
    h2 = orc.gen_haystack(0, 5 << 20, seed=0xAC02, lo=0x61, span=26)
    a2, o2 = build_pair(az, "standard", {"kind": "dfa"})
    assert_same(a2.find_overlapping_iter(h2, as_numpy=True), o2.find_overlapping_iter(h2, as_numpy=True), "host dense")
    # stream: 4 MiB reads, i.e. every feed is a large host chunk whose pieces are fed behind the copy
    got = triples(a.stream_find_iter(io.BytesIO(hay.tobytes()), chunk_bytes=(4 << 20) + 333))
    assert got == want_triples(orc.Oracle(pats), hay)
