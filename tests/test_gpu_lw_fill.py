"""-m gpu: the record fill from the LDS image of the LDS walk's one-row-per-state form (device/lds_walk.hip: k_lw_fill)
through the public calls -- overlapping searches of small pattern sets over match-dense input, synchronous and enqueue-only,
host and device output, shards and spans -- against the oracle's ordered stream."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
from gpu_util import assert_same, build_pair
from oracle import orc

pytestmark = pytest.mark.gpu


def sets():
    rng = np.random.default_rng(5)
    yield "one byte", [b"a"], np.frombuffer(b"a" * (3 << 20), dtype=np.uint8).copy()
    yield "nested + duplicates + empty", [b"", b"a", b"aa", b"aaa", b"a", b"ba", b"ab"], rng.choice(np.frombuffer(b"ab\n", dtype=np.uint8), size=1 << 20)
    alpha = np.frombuffer(b"etaoin shr", dtype=np.uint8)
    pats = [bytes(rng.choice(alpha, size=int(rng.integers(1, 4)))) for _ in range(48)]
    yield "48 short patterns", pats, rng.choice(np.concatenate([alpha, np.arange(256, dtype=np.uint8)]), size=2 << 20).astype(np.uint8)
    import corpora
    yield "prose", [b"the", b"he", b"e", b" ", b"Holmes", b"\r\n"], np.tile(corpora.haystack("sherlock.txt"), 4)


@pytest.mark.parametrize("engine", ["auto", "hot", "pf"])
def test_records_of_match_dense_small_sets(engine):
    for name, pats, hay in sets():
        if engine != "auto" and b"" in pats:
            continue   # (neither the LDS walk nor the prefix filter is offered for sets with an empty pattern)
        a, o = build_pair(pats, "standard", {"kind": "dfa"}, engine=engine)
        want = o.find_overlapping_iter(hay, as_numpy=True)
        assert len(want) > len(hay) // 16, name
        dev = torch.from_numpy(hay).cuda()
        assert_same(a.find_overlapping_iter(dev, as_numpy=True), want, f"{name} / {engine} / device haystack")
        assert_same(a.find_overlapping_iter(hay, as_numpy=True), want, f"{name} / {engine} / host haystack")
        # device-resident output through the *_device form, then a span with a shard inside it
        out = torch.empty(len(want) * 24 + 4096, dtype=torch.uint8, device="cuda")
        n, ok = a.overlapping_device(dev, out=out)
        assert ok and n == len(want)
        got = out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE)
        assert_same(got, want, f"{name} / {engine} / device output")
        lo, hi = 1000, len(hay) - 777
        sub = o.find_overlapping_iter(hay, span=(lo, hi), as_numpy=True)
        n, ok = a.overlapping_device(dev, span=(lo, hi), out=out)
        assert ok
        assert_same(out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE), sub, f"{name} / {engine} / span")


def test_enqueue_form_dense():
    name, pats, hay = next(iter(sets()))
    a, o = build_pair([b"a", b"aa"], "standard", {"kind": "dfa"})
    want = o.find_overlapping_iter(hay, as_numpy=True)
    dev = torch.from_numpy(hay).cuda()
    out = torch.empty(len(want) * 24 + 4096, dtype=torch.uint8, device="cuda")
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    for _ in range(3):   # (chunk counters -> scan -> fill: the form without an occurrence limit)
        a.overlapping_enqueue(dev, out, totals, classic=True)
    torch.cuda.synchronize()
    t = totals.cpu().numpy()
    assert int(t[0]) == len(want) and int(t[1]) == 0
    assert_same(out[: len(want) * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, "enqueue dense")
