"""-m gpu: the reference's own benchmark inputs (natural text searched for dictionary words: where a 4-byte prefix filter
behaves nothing like on random ASCII) and an adversarial haystack made of pattern prefixes -- every engine, and the
automatic choice between them, against the oracle's stream."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
import corpora
from gpu_util import assert_same, build_pair
from oracle import orc

pytestmark = pytest.mark.gpu


def dev(h):
    return torch.from_numpy(h).cuda()


@pytest.mark.parametrize("pats", ["words-100", "words-5000", "words-15000", "dictionary-15"])
@pytest.mark.parametrize("hay", ["sherlock.txt", "en-huge.txt"])
def test_natural_text_dictionary_patterns(pats, hay):
    ws = corpora.words(pats)
    h = corpora.haystack(hay, 24 << 20)
    o = orc.Oracle(ws, kind=orc.KIND_CNFA)
    want, want_hash = o.find_overlapping_parallel(h)
    d = dev(h)
    for engine in ("auto", "pf", "hot", "walk"):
        try:
            a = ac.AhoCorasick.builder().gpu_engine(engine).build(ws)   # default kind (contiguous NFA beyond 100 patterns)
            got = a.find_overlapping_iter(d, as_numpy=True)
        except RuntimeError as e:   # "requested engine is unavailable": the LDS walk needs the automaton to fit in LDS
            assert engine == "hot" and "invalid argument" in str(e), (engine, e)
            continue
        assert_same(got, want, f"{pats} on {hay}, engine {engine}")
        assert orc.hash_matches(got) == want_hash
    for mk in ("leftmost_first", "leftmost_longest"):
        a, o2 = build_pair(ws, mk)
        sub = h[: 4 << 20]
        assert_same(a.find_iter(dev(sub), as_numpy=True), o2.find_iter(sub, as_numpy=True), f"{pats} on {hay} find_iter {mk}")


@pytest.mark.parametrize("kind", ["prefix4", "prefix8", "whole"])
def test_adversarial_haystack_of_pattern_prefixes(kind):
    """Every position of the haystack begins a 4-byte (8-byte / whole) prefix of some pattern: the prefix filter's
    level 1 passes everything; results stay exact and the automatic engine choice must not be slower than the walk."""
    pats = orc.gen_patterns(1000, seed=0xAC01)
    rng = np.random.default_rng(3)
    k = {"prefix4": 4, "prefix8": 8, "whole": 99}[kind]
    pieces = [pats[int(i)][:k] for i in rng.integers(0, len(pats), size=(8 << 20) // 4)]
    h = np.frombuffer(b"".join(pieces), dtype=np.uint8)[: 8 << 20].copy()
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    want, want_hash = o.find_overlapping_parallel(h)
    d = dev(h)
    for engine in ("auto", "pf", "hot", "walk"):
        a, _ = build_pair(pats, "standard", {"kind": "dfa"}, engine=engine)
        prof = ac._lib.CProfile()
        got = a.find_overlapping_iter(d, as_numpy=True, profile=prof)
        assert_same(got, want, f"adversarial {kind}, engine {engine}")
        assert orc.hash_matches(got) == want_hash
        if engine == "auto":   # the filter's cost model hands this input to the LDS transition walk
            assert int(prof.routed) == 1 and int(prof.engine_used) == 3, (int(prof.routed), int(prof.engine_used))
        if engine == "pf":     # an explicitly requested engine is kept
            assert int(prof.routed) == 0 and int(prof.engine_used) == 4
    # the enqueue-only form reports an abandoned scan as an event overflow: "repeat with the synchronous call"
    a, _ = build_pair(pats, "standard", {"kind": "dfa"})
    out = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    a.overlapping_enqueue(d, out, totals)
    torch.cuda.synchronize()
    assert int(totals.cpu().numpy().view(np.uint64)[1]) > a.ENQUEUE_MAX_EVENTS
    # ... and the next call on the same stream starts from re-armed counters
    sparse = orc.gen_haystack(0, 1 << 22, seed=0xAC02)
    a.overlapping_enqueue(dev(sparse), out, totals)
    torch.cuda.synchronize()
    t = totals.cpu().numpy()
    w2 = o.find_overlapping_iter(sparse, as_numpy=True)
    assert int(t[0]) == len(w2)
    assert_same(out[: len(w2) * 24].cpu().numpy().view(ac.MATCH_DTYPE), w2, "enqueue after an abandoned scan")


def test_routing_leaves_ordinary_inputs_with_the_filter():
    pats = orc.gen_patterns(1000, seed=0xAC01)
    prof = ac._lib.CProfile()
    for lo, span in ((0x20, 95), (0x61, 26)):
        p = orc.gen_patterns(1000, seed=0xAC01, lo=lo, span=span)
        h = orc.gen_haystack(0, 32 << 20, seed=0xAC02, lo=lo, span=span)
        a, o = build_pair(p, "standard", {"kind": "dfa"})
        got = a.find_overlapping_iter(dev(h), as_numpy=True, profile=prof)
        assert int(prof.routed) == 0 and int(prof.engine_used) == 4
        want, _ = o.find_overlapping_parallel(h)
        assert_same(got, want, f"alphabet {span}")
    text = corpora.haystack("sherlock.txt", 32 << 20)
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build(pats)
    a.find_overlapping_iter(dev(text), as_numpy=True, profile=prof)
    assert int(prof.routed) == 0 and int(prof.engine_used) == 4


@pytest.mark.parametrize("hay,pats", [("sherlock.txt", "words-5000"), ("en-huge.txt", "words-15000")])
def test_natural_text_probe_and_enqueue_form(hay, pats):
    """The workloads of the reference's curated/sherlock benchmarks through the pipelined (enqueue-only) form.  The first
    synchronous call pays for an abandoned two-type pass and remembers it; from then on the probe (k_pf_probe) decides on
    the device, both filters are enqueued gated on its word, and more than 16 384 occurrences are ordered by the bucket
    pass behind the scan -- no host decision, records bit-exact, totals[1] == 0."""
    ws = corpora.words(pats)
    h = corpora.haystack(hay, 96 << 20)
    o = orc.Oracle(ws, kind=orc.KIND_CNFA)
    want, want_hash = o.find_overlapping_parallel(h)
    assert len(want) > 50000
    d = dev(h)
    a = ac.AhoCorasick.builder().build(ws)
    prof = ac._lib.CProfile()
    got = a.find_overlapping_iter(d, as_numpy=True, profile=prof)
    assert_same(got, want, "synchronous, first call")
    first_routed = int(prof.routed)
    got = a.find_overlapping_iter(d, as_numpy=True, profile=prof)      # (the probe instead of an abandoned pass)
    assert_same(got, want, "synchronous, probed")
    assert int(prof.routed) == first_routed
    out = torch.full((len(want) * 24 + 4096,), 0xEE, dtype=torch.uint8, device="cuda")
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    for _ in range(3):
        a.overlapping_enqueue(d, out, totals)
    torch.cuda.synchronize()
    t = totals.cpu().numpy().view(np.uint64)
    assert int(t[0]) == len(want) and int(t[1]) == 0, t
    got = out[: len(want) * 24].cpu().numpy().view(ac.MATCH_DTYPE)
    assert_same(got, want, f"{pats} on {hay}, enqueue form")
    assert orc.hash_matches(got) == want_hash
    # the same automaton on harmless input afterwards: the probe says "filter", the hint runs out, results stay exact
    rnd = orc.gen_haystack(0, 32 << 20, seed=0xAC02)
    w2 = o.find_overlapping_iter(rnd, as_numpy=True)
    dr = dev(rnd)
    for _ in range(10):
        assert_same(a.find_overlapping_iter(dr, as_numpy=True), w2, "random text after natural text")
    a.overlapping_enqueue(dr, out, totals)
    torch.cuda.synchronize()
    t = totals.cpu().numpy().view(np.uint64)
    assert int(t[0]) == len(w2) and int(t[1]) <= a.ENQUEUE_MAX_EVENTS
    assert_same(out[: len(w2) * 24].cpu().numpy().view(ac.MATCH_DTYPE), w2, "random text, enqueue form")
