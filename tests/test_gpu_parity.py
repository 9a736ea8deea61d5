"""-m gpu: HIP path vs the CPU oracle on seeded synthetic inputs -- bit-exact ordered triple lists."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
from gpu_util import assert_same, build_pair, plant
from oracle import orc

pytestmark = pytest.mark.gpu


def dev(h):
    return torch.from_numpy(h).cuda()


@pytest.fixture(scope="module")
def c2_patterns():
    return orc.gen_patterns(1000, seed=0xAC01)


@pytest.mark.parametrize("engine", ["walk", "hot", "pf"])
@pytest.mark.parametrize("chunk", [64, 256, 4096])
def test_c2_overlapping_with_planted_seams(c2_patterns, engine, chunk):
    n = 1 << 21
    hay = orc.gen_haystack(0, n, seed=0xAC02)
    # plant occurrences straddling every kind of lane-chunk / wave / block seam
    pos = [chunk * k - d for k, d in zip((1, 3, 5, 63, 64, 65, 127, 255, 256, 257, 300, 511, 512, 1000, 4095, 4096),
                                         (0, 1, 3, 7, 15, 16, 2, 5, 9, 11, 4, 8, 12, 13, 14, 6))]
    pos += [0, n - 16, n - 4] + [65536 * k + 17 * k for k in range(1, 30)]
    plant(hay, c2_patterns[:50], pos)
    a, o = build_pair(c2_patterns, "standard", {"kind": "dfa"}, chunk=chunk, engine=engine)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > 30
    got = a.find_overlapping_iter(dev(hay), as_numpy=True)
    assert_same(got, want, f"c2 {engine} chunk={chunk}")


@pytest.mark.parametrize("kind,engine", [("dfa", "walk"), ("dfa", "hot"), ("dfa", "pf"), ("cnfa", "auto"),
                                         ("nnfa", "auto")])
def test_dense_matches_small_alphabet(kind, engine):
    """a-z alphabet: ~10^5 matches per MiB, exercises count/scan/fill with every chunk non-empty."""
    pats = [p[:3] for p in orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)]  # 3-byte patterns, with duplicates
    hay = orc.gen_haystack(0, 1 << 20, seed=0xAC02, lo=0x61, span=26)
    a, o = build_pair(pats, "standard", {"kind": kind}, chunk=256, engine=engine)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > 1000
    assert_same(a.find_overlapping_iter(dev(hay), as_numpy=True), want, f"{kind} {engine}")


@pytest.mark.parametrize("chunk", [0, 4096])
def test_small_alphabet_wide_row_index(chunk):
    """1 000 a-z patterns on the LDS walk: 27 classes, 625 rows -- the 10 | 6 | 16 handle layout (host model:
    tests/test_lw_tables.py).  8 MiB so that most wave tasks are interior ones (fast step, next-task prefetch) and the
    first / last regions take the edge walk."""
    pats = orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)
    hay = orc.gen_haystack(0, 8 << 20, seed=0xAC02, lo=0x61, span=26)
    a, o = build_pair(pats, "standard", {"kind": "dfa"}, chunk=chunk, engine="hot")
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > 1000
    assert_same(a.find_overlapping_iter(dev(hay), as_numpy=True), want, f"a-z hot chunk={chunk}")
    # a span that starts and ends inside the buffer, misaligned
    got = a.find_overlapping_iter(ac.Input(dev(hay)).span((12345, (8 << 20) - 777)), as_numpy=True)
    assert_same(got, o.find_overlapping_iter(hay, as_numpy=True, span=(12345, (8 << 20) - 777)), "a-z hot span")


@pytest.mark.parametrize("engine", ["walk", "hot", "pf"])
def test_short_and_nested_patterns(engine):
    """1- and 2-byte patterns, nested/duplicate patterns, case-insensitive: the 'always verify' arms."""
    pats = [b"a", b"ab", b"abc", b"bc", b"c", b"abc", b"bca", b"cabcab", b"zz", b"z", b"q"]
    rng = np.random.default_rng(5)
    hay = rng.integers(97, 100, size=1 << 16, dtype=np.uint8)
    hay[1000:1010] = ord("z")
    hay[5000] = ord("q")
    a, o = build_pair(pats, "standard", {"kind": "dfa"}, chunk=64, engine=engine)
    assert_same(a.find_overlapping_iter(dev(hay), as_numpy=True), o.find_overlapping_iter(hay, as_numpy=True), engine)
    ci = [b"Foo", b"fOOb", b"BAR", b"barf", b"o"]
    hay2 = np.frombuffer((b"xfoobarFOOBARf--fooBARFoo" * 3000), dtype=np.uint8).copy()
    a, o = build_pair(ci, "standard", {"kind": "dfa", "ascii_case_insensitive": True}, chunk=128, engine=engine)
    assert_same(a.find_overlapping_iter(dev(hay2), as_numpy=True), o.find_overlapping_iter(hay2, as_numpy=True),
                f"casei {engine}")


def test_host_haystack_and_subspan(c2_patterns):
    hay = orc.gen_haystack(0, 1 << 18, seed=7)
    plant(hay, c2_patterns[:20], [100, 1000, 4090, 4100, 65530, 100000])
    a, o = build_pair(c2_patterns, "standard", {"kind": "dfa"}, chunk=128)   # auto engine == prefix filter
    for span in [(0, len(hay)), (1, 4097), (95, 105), (4095, 4096), (4096, 4096), (100, 100), (101, 116), (990, 1016)]:
        want = o.find_overlapping_iter(hay, span=span, as_numpy=True)
        got = a.find_overlapping_iter(ac.Input(hay).range(*span), as_numpy=True)      # host numpy -> staged
        assert_same(got, want, f"host span={span}")
        got = a.find_overlapping_iter(ac.Input(dev(hay)).range(*span), as_numpy=True)  # device resident
        assert_same(got, want, f"dev span={span}")
    # a pattern occurrence that begins before span.start is NOT reported (cold start)
    p = c2_patterns[0]
    h = np.frombuffer(b"zz" + p + b"zz", dtype=np.uint8).copy()
    assert len(a.find_overlapping_iter(ac.Input(h).range(3, len(h)), as_numpy=True)) == \
        len(o.find_overlapping_iter(h, span=(3, len(h))))


@pytest.mark.parametrize("mis", [0, 1, 5, 15, 17])
def test_misaligned_device_pointer(c2_patterns, mis):
    n = 70000
    hay = orc.gen_haystack(3, n, seed=11)
    plant(hay, c2_patterns[:10], [0, 60, 64 - mis, 128 - mis, 4096 - mis - 3, n - 8])
    big = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    big[mis:mis + n] = torch.from_numpy(hay).cuda()
    for engine in ("walk", "hot", "pf"):
        a, o = build_pair(c2_patterns, "standard", {"kind": "dfa"}, chunk=64, engine=engine)
        assert_same(a.find_overlapping_iter(big[mis:mis + n], as_numpy=True),
                    o.find_overlapping_iter(hay, as_numpy=True), f"mis={mis} {engine}")


def test_shards_concatenate_to_the_full_stream(c2_patterns):
    """acgpu_find_overlapping_shard: consecutive shards with max_pattern_len-1 warm-up tile the full result."""
    n = 1 << 20
    hay = orc.gen_haystack(0, n, seed=0xAC02, lo=0x61, span=26)
    pats = orc.gen_patterns(300, seed=5, lo=0x61, span=26)
    d = dev(hay)
    want = None
    for engine in ("walk", "pf"):
        a, o = build_pair(pats, "standard", {"kind": "dfa"}, chunk=512, engine=engine)
        want = o.find_overlapping_iter(hay, span=(10, n - 3), as_numpy=True)
        parts = [a.find_overlapping_shard(ac.Input(d).range(10, n - 3), sb, se)
                 for sb, se in zip([10, 4097, 70001], [4097, 70001, n - 3])]
        assert_same(np.concatenate(parts), want, f"shards {engine}")
    for cuts in ([10, n - 3], [10, 11, 5000, 5001, 5017, 300000, n - 3], [10, 10, 70000, 70000, n - 3]):
        parts = []
        for sb, se in zip(cuts[:-1], cuts[1:]):
            parts.append(a.find_overlapping_shard(ac.Input(d).range(10, n - 3), sb, se))
        assert_same(np.concatenate(parts), want, f"cuts={cuts}")


def test_empty_patterns_and_long_patterns():
    # empty patterns: every position matches; plus the start-state matches at span.start
    for pats in ([b"", b"a", b"ba"], [b"", b"b", b"ab"], [b"a", b"", b""], [b""]):
        hay = np.frombuffer(b"abbaababbab" * 40, dtype=np.uint8).copy()
        for kind in ("dfa", "cnfa"):
            a, o = build_pair(pats, "standard", {"kind": kind}, chunk=64)
            for span in [(0, len(hay)), (5, 200), (64, 64), (440, 440)]:
                assert_same(a.find_overlapping_iter(ac.Input(dev(hay)).range(*span), as_numpy=True),
                            o.find_overlapping_iter(hay, span=span, as_numpy=True), f"{pats} {kind} {span}")
    # patterns longer than one 64-byte tile: multi-tile warm-up halo
    rng = np.random.default_rng(3)
    longp = [bytes(rng.integers(97, 100, size=L, dtype=np.uint8)) for L in (70, 130, 200, 3, 65)]
    hay = rng.integers(97, 100, size=1 << 16, dtype=np.uint8)
    plant(hay, longp, [0, 30, 64 * 7 - 60, 64 * 20 - 1, 64 * 33 - 199, 5000, 60000])
    for engine in ("walk", "hot", "pf"):
        a, o = build_pair(longp, "standard", {"kind": "dfa"}, chunk=64, engine=engine)
        assert_same(a.find_overlapping_iter(dev(hay), as_numpy=True), o.find_overlapping_iter(hay, as_numpy=True),
                    f"long {engine}")


@pytest.mark.parametrize("engine", ["walk", "auto"])
def test_c4_contiguous_nfa(engine):
    """BASELINE config 4: 100 000 patterns, AhoCorasickKind::ContiguousNFA.  "walk" = the contiguous-NFA failure-link
    walk itself (src/nfa/contiguous.rs:186-247) on the device; "auto" = the full DFA derived from the same
    noncontiguous NFA in HBM (rows filled on the device).  Both must reproduce the oracle's stream."""
    pats = orc.gen_patterns(100000, seed=0xAC04)
    hay = orc.gen_haystack(0, 1 << 21, seed=0xAC02)
    plant(hay, pats[:64], [4096 * k - 5 for k in range(1, 60)])
    a, o = build_pair(pats, "standard", {"kind": "cnfa"}, engine=engine)
    assert a.kind() == ac.AhoCorasickKind.ContiguousNFA
    assert_same(a.find_overlapping_iter(dev(hay), as_numpy=True), o.find_overlapping_iter(hay, as_numpy=True), f"c4 {engine}")


def test_c5_leftmost_first_case_insensitive(c2_patterns):
    """BASELINE config 5: ascii_case_insensitive + LeftmostFirst, non-overlapping find_iter."""
    n = 1 << 18
    hay = orc.gen_haystack(0, n, seed=0xAC05)
    planted = [p.swapcase() for p in c2_patterns[:40]]
    plant(hay, planted, list(range(500, n - 100, 3001)))
    for kind in ("dfa", "cnfa"):
        a, o = build_pair(c2_patterns, "leftmost_first", {"kind": kind, "ascii_case_insensitive": True})
        want = o.find_iter(hay, as_numpy=True)
        assert len(want) > 40
        assert_same(a.find_iter(dev(hay), as_numpy=True), want, f"c5 {kind}")


def test_gen_haystack_matches_cpu_generator():
    t = torch.empty(100003, dtype=torch.uint8, device="cuda")
    ac.gen_haystack(t, offset=12345, seed=0xAC02)
    assert np.array_equal(t.cpu().numpy(), orc.gen_haystack(12345, 100003, seed=0xAC02))
    t2 = t[1:5000]
    ac.gen_haystack(t2, offset=9, seed=3, lo=0x61, span=26)
    assert np.array_equal(t2.cpu().numpy(), orc.gen_haystack(9, 4999, seed=3, lo=0x61, span=26))


@pytest.mark.parametrize("mk", ["standard", "leftmost_first", "leftmost_longest"])
@pytest.mark.parametrize("kind", ["dfa", "cnfa", "nnfa", None])
def test_find_iter_parallel_path(c2_patterns, mk, kind):
    """find_iter through the parallel path (occurrence stream + device selection), incl. sub-spans."""
    n = 1 << 20
    hay = orc.gen_haystack(0, n, seed=0xAC06, lo=0x61, span=26)
    pats = [p[:k] for p, k in zip(orc.gen_patterns(400, seed=9, lo=0x61, span=26), [2, 3, 3, 4, 5] * 80)]
    a, o = build_pair(pats, mk, {"kind": kind}, chunk=256)
    want = o.find_iter(hay, as_numpy=True)
    assert len(want) > 1000
    assert_same(a.find_iter(dev(hay), as_numpy=True), want, f"{mk} {kind}")
    for span in [(1, n - 1), (777, 77777), (5000, 5003), (100, 100)]:
        assert_same(a.find_iter(ac.Input(dev(hay)).range(*span), as_numpy=True),
                    o.find_iter(hay, span=span, as_numpy=True), f"{mk} {kind} {span}")
    # case-insensitive variant
    a, o = build_pair(c2_patterns, mk, {"kind": kind, "ascii_case_insensitive": True})
    hay2 = orc.gen_haystack(0, 1 << 19, seed=0xAC07)
    plant(hay2, [p.swapcase() for p in c2_patterns[:60]], list(range(100, (1 << 19) - 100, 2999)))
    assert_same(a.find_iter(dev(hay2), as_numpy=True), o.find_iter(hay2, as_numpy=True), f"casei {mk} {kind}")


@pytest.mark.parametrize("seed", range(8))
def test_prefix_filter_randomized(seed):
    """Randomised differential test of the prefix-filter count engine: tiny alphabets (dense matches, nested and
    1-3 byte patterns exercise the wildcarded table entries), random spans of either parity, misaligned bases."""
    rng = np.random.default_rng(1000 + seed)
    for case in range(12):
        sigma = int(rng.integers(2, 7))
        npat = int(rng.integers(1, 40))
        pats = [bytes(rng.integers(0x61, 0x61 + sigma, size=int(rng.integers(1, 8)), dtype=np.uint8)) for _ in range(npat)]
        n = int(rng.integers(0, 6000))
        mis = int(rng.integers(0, 16))
        buf = np.zeros(n + 32, dtype=np.uint8)
        buf[mis:mis + n] = rng.integers(0x61, 0x61 + sigma + int(rng.integers(0, 2)), size=n, dtype=np.uint8)
        hay = buf[mis:mis + n]
        chunk = int(rng.choice([64, 128, 1024]))
        a, o = build_pair(pats, "standard", {"kind": "dfa"}, chunk=chunk, engine="pf")
        dbuf = dev(buf)
        dh = dbuf[mis:mis + n]
        ctx = f"seed={seed} case={case} sigma={sigma} npat={npat} n={n} mis={mis} chunk={chunk}"
        assert_same(a.find_overlapping_iter(dh, as_numpy=True), o.find_overlapping_iter(hay, as_numpy=True), ctx)
        if n:
            s0 = int(rng.integers(0, n)); s1 = int(rng.integers(s0, n + 1))
            assert_same(a.find_overlapping_iter(ac.Input(dh).range(s0, s1), as_numpy=True),
                        o.find_overlapping_iter(hay, span=(s0, s1), as_numpy=True), ctx + f" span=({s0},{s1})")


@pytest.mark.parametrize("engine", ["pf", "hot", "walk"])
def test_saturated_haystack(engine):
    """Worst case for the filter engines: every position starts several pattern occurrences (queues, level 3 and the
    fill's re-walk path all saturate); 1.5 M+ matches, exact stream."""
    pats = [b"abab", b"ab", b"b", b"abababab", b"ba", b"bab"]
    hay = np.frombuffer(b"ab" * (1 << 19), dtype=np.uint8).copy()
    a, o = build_pair(pats, "standard", {"kind": "dfa"}, engine=engine)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > 1_500_000
    assert_same(a.find_overlapping_iter(dev(hay), as_numpy=True), want, f"saturated {engine}")
    # and the non-overlapping selection over that stream
    for mk in ("standard", "leftmost_first", "leftmost_longest"):
        a2, o2 = build_pair(pats, mk, {"kind": "dfa"}, engine="auto" if engine == "pf" else engine)
        assert_same(a2.find_iter(dev(hay), as_numpy=True), o2.find_iter(hay, as_numpy=True), f"saturated {mk} {engine}")


def test_prefix_filter_wide_alphabets():
    """The prefix filter indexes its bigram table by dense codes of the bytes that start patterns, so byte ranges far
    wider than printable ASCII (UTF-8 text, sparse binary alphabets) keep the fast engine."""
    rng = np.random.default_rng(33)
    words = ["naïve", "café", "€uro", "smörgåsbord", "日本語", "テキスト", "über", "résumé", "Ωmega", "zürich", "plain", "ascii"]
    pats = [w.encode() for w in words] + [bytes([b, 255 - b, 7, b ^ 0x55]) for b in range(0, 250, 5)]
    hay = rng.integers(0, 256, size=1 << 20, dtype=np.uint8)
    plant(hay, pats, list(range(1000, (1 << 20) - 100, 4099)))
    a, o = build_pair(pats, "standard", {"kind": "dfa"}, engine="pf")     # raises if the filter is unavailable
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > 200
    assert_same(a.find_overlapping_iter(dev(hay), as_numpy=True), want, "wide alphabet pf")
    text = ("Ein naïve café in zürich: résumé über 日本語 テキスト, €uro Ωmega plain ascii. " * 2000).encode()
    t = np.frombuffer(text, dtype=np.uint8)
    assert_same(a.find_overlapping_iter(dev(t.copy()), as_numpy=True), o.find_overlapping_iter(t, as_numpy=True), "utf-8 text")


@pytest.mark.parametrize("npat", [5000, 12000, 30000])
def test_prefix_filter_mid_size_pattern_sets(npat):
    """Pattern sets beyond 32k automaton states keep the prefix filter (32-bit trie table); the Bloom table passes
    more, the result stays exact.  30000 patterns: the second table is keyed by true starts (HotTables::pf_exact2)."""
    pats = orc.gen_patterns(npat, seed=0xAC05)
    hay = orc.gen_haystack(0, 1 << 22, seed=0xAC02)
    plant(hay, pats[::37], [8191 * k - 3 for k in range(1, 500)])
    # 1-, 2- and 3-byte patterns take the wildcard arms of both table constructions
    pats = pats + [b"\x01", b"\x02\x03", b"\x04\x05\x06"]
    hay[12345] = 1
    hay[70001:70003] = (2, 3)
    hay[(1 << 22) - 3:] = (4, 5, 6)
    a, o = build_pair(pats, "standard", {"kind": "dfa"}, engine="pf")
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > 400
    assert_same(a.find_overlapping_iter(dev(hay), as_numpy=True), want, f"pf {npat} patterns")
    # default automaton kind (contiguous NFA beyond 100 patterns) reaches the same engine through the derived DFA
    a2, _ = build_pair(pats, "standard", {})
    assert a2.kind() == ac.AhoCorasickKind.ContiguousNFA
    assert_same(a2.find_overlapping_iter(dev(hay), as_numpy=True), want, f"auto kind {npat} patterns")


@pytest.mark.parametrize("npat,gate", [(300, 0), (5000, 0), (30000, 0), (100000, 0), (5000, 1), (100000, 1)])
def test_large_set_filter_with_verifier_wavefronts(npat, gate):
    """pfx_scan.hip (4-byte-key blocked Bloom table, producer / verifier wavefronts) forced for every set size it can
    serve: sparse and dense inputs, sub-spans of every alignment, shards, the non-overlapping iterator on top, and a
    haystack made of pattern prefixes (rings full, producers waiting for their verifier).  gate = 1: with the
    exact-prefix bit table in front of the hash map and the map lookup in the second pass (off by default)."""
    forms = {"pfx_min_patterns": 1, "pfx_gate": gate}   # engine variants of the automata under test
    pats = orc.gen_patterns(npat, seed=0xAC05)
    n = 6 << 20
    hay = orc.gen_haystack(0, n, seed=0xAC02)
    plant(hay, pats[::max(1, npat // 200)], [8191 * k - 3 for k in range(1, 700)] + [0, n - 4, n - 16])
    a, o = build_pair(pats, "standard", {"kind": "dfa"} if npat <= 30000 else {"kind": "cnfa"}, engine="pf", variants=forms)
    want, want_hash = o.find_overlapping_parallel(hay)
    assert len(want) > 500
    d = dev(hay)
    prof = ac._lib.CProfile()
    got = a.find_overlapping_iter(d, as_numpy=True, profile=prof)
    assert int(prof.engine_used) == 4
    assert_same(got, want, f"pfx {npat}")
    rng = np.random.default_rng(npat)
    for _ in range(4):
        s0 = int(rng.integers(0, n // 2)); s1 = int(rng.integers(s0, n + 1))
        sub = want[(want["start"] >= s0) & (want["end"] <= s1)]
        assert_same(a.find_overlapping_iter(ac.Input(d).range(s0, s1), as_numpy=True), sub, f"pfx {npat} span ({s0},{s1})")
    mid = n // 2 + 13
    parts = [a.find_overlapping_shard(ac.Input(d), 0, mid), a.find_overlapping_shard(ac.Input(d), mid, n)]
    assert_same(np.concatenate(parts), want, f"pfx {npat} shards")
    # dense: haystack of pattern prefixes and whole patterns
    rng = np.random.default_rng(7)
    pieces = [pats[int(i)][: int(k)] for i, k in zip(rng.integers(0, len(pats), size=200000), rng.integers(4, 17, size=200000))]
    h2 = np.frombuffer(b"".join(pieces), dtype=np.uint8)[: 1 << 20].copy()
    want2, _ = o.find_overlapping_parallel(h2)
    assert len(want2) > 50000
    assert_same(a.find_overlapping_iter(dev(h2), as_numpy=True), want2, f"pfx {npat} dense")
    out = torch.zeros(len(want2) * 24 + 4096, dtype=torch.uint8, device="cuda")   # chunk-counter form (count -> scan -> fill)
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    a.overlapping_enqueue(dev(h2), out, totals, classic=True)
    torch.cuda.synchronize()
    assert int(totals.cpu().numpy()[0]) == len(want2)
    assert_same(out[: len(want2) * 24].cpu().numpy().view(ac.MATCH_DTYPE), want2, f"pfx {npat} dense, chunk counters")
    if npat <= 30000:
        lf, olf = build_pair(pats, "leftmost_first", {"kind": "dfa"}, variants=forms)
        assert_same(lf.find_iter(dev(h2), as_numpy=True), olf.find_iter(h2, as_numpy=True), f"pfx {npat} find_iter")


@pytest.mark.parametrize("minlen,tails,roles,x2", [(5, 1, 12, 1), (6, 1, 12, 1), (7, 1, 12, 1), (8, 1, 12, 1), (9, 1, 12, 1), (11, 1, 12, 1),
                                                   (8, 0, 12, 1), (11, 0, 12, 1), (8, 1, 14, 1), (11, 1, 14, 1), (9, 1, 12, 0), (11, 1, 12, 0)])
def test_large_set_filter_long_prefix_level2(minlen, tails, roles, x2):
    """pfx_scan.hip with the long-prefix map (HotTables::pfx_map8): when the shortest pattern has 5..8+ bytes, level 2
    compares min(8, shortest) bytes exactly (bytes 4.. fetched from the haystack by the verifier).  Patterns sharing
    4..7-byte prefixes, occurrences touching both ends of the span, spans ending inside a prefix, shards, a haystack
    made of pattern prefixes (hit-dense: level 3 goes to the second pass).  tails = 0: without the chain-tail records
    behind the map (level 3 walks the trie from the prefix node for every hit; with them only where the trie branches,
    a pattern ends inside the chain, or the span ends within 16 bytes)."""
    # engine variants: producers of the 16 wavefronts under the long-key level 1 (12 = default); x2 = 0: from 9-byte patterns
    # on that level 1 probes every other position (one hash for two starts) -- off here
    forms = {"pfx_min_patterns": 1, "pfx_tails": tails, "pfx_key8_roles": roles, "pfx_key8_x2": x2}
    rng = np.random.default_rng(minlen)
    base = orc.gen_patterns(3000, seed=0xAC06 + minlen, lo=0x61, span=26)
    pats = []
    for i, p in enumerate(base):   # lengths minlen..minlen+9, every fifth pattern shares a 4..7-byte prefix with its predecessor
        ln = minlen + int(rng.integers(0, 10))
        body = (p * 4)[:ln]
        if i % 5 == 4 and pats:
            k = int(rng.integers(4, min(8, minlen) + 1))
            body = pats[-1][:k] + body[k:]
        pats.append(bytes(body))
    n = 3 << 20
    hay = orc.gen_haystack(0, n, seed=0xAC02, lo=0x61, span=26)
    plant(hay, pats[::15], [8191 * k - 3 for k in range(1, 350)])
    for p, at in ((pats[0], 0), (pats[1], n - len(pats[1])), (pats[2][: minlen - 1], n - 40)):   # both ends; a bare prefix
        hay[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
    a, o = build_pair(pats, "standard", {"kind": "dfa"}, engine="pf", variants=forms)
    want, _ = o.find_overlapping_parallel(hay)
    assert len(want) > 300
    d = dev(hay)
    prof = ac._lib.CProfile()
    assert_same(a.find_overlapping_iter(d, as_numpy=True, profile=prof), want, f"long prefix {minlen}")
    assert int(prof.engine_used) == 4
    last = int(want["end"][-1])
    for s0, s1 in [(0, last - 1), (1, last - 3), (5, last), (0, n - len(pats[1]) + 6), (0, n - len(pats[1]) + 4)] + \
                  [tuple(sorted(int(x) for x in rng.integers(0, n, size=2))) for _ in range(3)]:
        sub = want[(want["start"] >= s0) & (want["end"] <= s1)]
        assert_same(a.find_overlapping_iter(ac.Input(d).range(s0, s1), as_numpy=True), sub, f"long prefix {minlen} span ({s0},{s1})")
    mid = n // 2 + 5
    parts = [a.find_overlapping_shard(ac.Input(d), 0, mid), a.find_overlapping_shard(ac.Input(d), mid, n)]
    assert_same(np.concatenate(parts), want, f"long prefix {minlen} shards")
    pieces = [pats[int(i)][: int(k)] for i, k in zip(rng.integers(0, len(pats), size=150000), rng.integers(4, minlen + 10, size=150000))]
    h2 = np.frombuffer(b"".join(pieces), dtype=np.uint8)[: 1 << 20].copy()
    want2, _ = o.find_overlapping_parallel(h2)
    assert len(want2) > 10000
    assert_same(a.find_overlapping_iter(dev(h2), as_numpy=True), want2, f"long prefix {minlen} dense")
    # chunk-counter form of the same kernels (count -> scan -> fill; the second pass credits chunk counters instead of
    # recording events): through the enqueue-only call with ACGPU_ENQUEUE_CLASSIC, sparse and hit-dense input
    for name, hh, ww in (("sparse", hay, want), ("dense", h2, want2)):
        out = torch.zeros(len(ww) * 24 + 4096, dtype=torch.uint8, device="cuda")
        totals = torch.zeros(2, dtype=torch.int64, device="cuda")
        a.overlapping_enqueue(dev(hh), out, totals, classic=True)
        torch.cuda.synchronize()
        assert int(totals.cpu().numpy()[0]) == len(ww), f"long prefix {minlen} counters {name}"
        assert_same(out[: len(ww) * 24].cpu().numpy().view(ac.MATCH_DTYPE), ww, f"long prefix {minlen} counters {name}")
    # spans that end right behind a shortest pattern: its start is closer than eight bytes to the end of the span (round 5: the
    # long-key level 1 for prefixes of 5..7 bytes read the key of such a start as zero; the fuzzer found it)
    shortest = min(pats, key=len)
    for at in (1000, 70_001, n - len(shortest) - 3):
        h3 = hay.copy()
        h3[at:at + len(shortest)] = np.frombuffer(shortest, dtype=np.uint8)
        for end in (at + len(shortest), at + len(shortest) + 1, at + len(shortest) + 7):
            end = min(end, n)
            w3 = o.find_overlapping_iter(h3, span=(max(0, at - 40), end), as_numpy=True)
            assert len(w3) >= 1
            assert_same(a.find_overlapping_iter(ac.Input(dev(h3)).range(max(0, at - 40), end), as_numpy=True), w3, f"long prefix {minlen} span ends at {end - at - len(shortest)} behind a shortest pattern")
    # a span shorter than the prefix, and one exactly as long as a pattern
    assert len(a.find_overlapping_iter(ac.Input(d).range(0, minlen - 1), as_numpy=True)) == 0
    assert_same(a.find_overlapping_iter(ac.Input(d).range(0, len(pats[0])), as_numpy=True),
                want[want["end"] <= len(pats[0])], f"long prefix {minlen} span = first pattern")


def test_prefix_filter_many_tasks_random_spans(c2_patterns):
    """Haystacks large enough that every wavefront of the prefix filter runs several tasks (so its loads are carried
    from one task into the next and the last tasks run guarded), searched over random spans of every alignment;
    the reference-faithful walk engine on the same device is the checker (itself pinned to the oracle above)."""
    rng = np.random.default_rng(77)
    n = 600 << 20
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    ac.gen_haystack(buf, offset=0, seed=0xAC0A)
    task = 40 * 1008
    for j in range(400):   # occurrences around task / row boundaries all over the buffer
        p = c2_patterns[(11 * j) % len(c2_patterns)]
        pos = int(rng.integers(1, n // task - 1)) * task + int(rng.integers(-20, 20))
        buf[pos:pos + len(p)] = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
    pf, _ = build_pair(c2_patterns, "standard", {"kind": "dfa"}, engine="pf")
    walk, _ = build_pair(c2_patterns, "standard", {"kind": "dfa"}, chunk=4096, engine="walk")
    for case in range(8):
        mis = int(rng.integers(0, 16))
        lo = int(rng.integers(0, 4 * task))
        hi = n - int(rng.integers(0, 4 * task))
        if case == 0:
            mis, lo, hi = 0, 0, n
        view = buf[mis:mis + n]
        inp = ac.Input(view).range(lo, hi - mis)
        got = pf.find_overlapping_iter(inp, as_numpy=True)
        want = walk.find_overlapping_iter(inp, as_numpy=True)
        assert len(want) > 400
        assert_same(got, want, f"case {case} mis={mis} span=({lo},{hi - mis})")


def test_large_result_sets_sorted_event_mode():
    """More occurrences than the all-pairs rank orders (16384): the level-3 events are ordered by the device radix sort
    (event_sort.hip).  Covers the first call (all-pairs attempt overflows, exact count sizes the sort), later calls
    (density known: sorted mode directly), device-resident output, a too-small buffer, the parallel find_iter on the
    same occurrence stream, and a jump in density between two haystacks (sort buffer too small -> classic pipeline)."""
    pats = orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)
    n = 160 << 20
    hay = orc.gen_haystack(0, n, seed=0xAC02, lo=0x61, span=26)
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > 20000
    d = dev(hay)
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_engine("pf").build(pats)
    prof = ac._lib.CProfile()
    for call in range(3):
        assert_same(a.find_overlapping_iter(d, as_numpy=True), want, f"host output, call {call}")
    out = torch.zeros(len(want) * 24, dtype=torch.uint8, device="cuda")
    m, ok = a.overlapping_device(d, out=out, profile=prof)
    assert ok and m == len(want) and prof.ms_compact == 0.0   # sorted mode reports no scan leg
    assert_same(out.cpu().numpy().view(ac.MATCH_DTYPE), want, "device output")
    small = torch.zeros(1000 * 24, dtype=torch.uint8, device="cuda")
    m, ok = a.overlapping_device(d, out=small)
    assert not ok and m == len(want)
    m, _ = a.overlapping_device(d, out=None)
    assert m == len(want)
    # sub-span + shard of the same haystack (seam ownership is by `end`)
    lo, hi = (n // 3) | 5, (2 * n // 3) | 9
    sub = o.find_overlapping_iter(hay[:hi], as_numpy=True)
    sub = sub[sub["end"] > lo]
    got = a.find_overlapping_iter(ac.Input(d).range(0, hi), as_numpy=True)
    assert_same(got[got["end"] > lo], sub, "span")
    # the non-overlapping iterator consumes the same ordered occurrence stream on the device
    lf = ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).kind(ac.AhoCorasickKind.DFA).build(pats)
    olf = orc.Oracle(pats, match_kind=1, kind=orc.KIND_DFA)
    for call in range(2):
        assert_same(lf.find_iter(d, as_numpy=True), olf.find_iter(hay, as_numpy=True), f"find_iter call {call}")
    # density jump: sparse haystack first (density ~ 0), then one 50x denser than the sort buffer was sized for
    b = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_engine("pf").build(pats)
    sparse = orc.gen_haystack(0, 1 << 20, seed=7)
    assert_same(b.find_overlapping_iter(dev(sparse), as_numpy=True), orc.Oracle(pats, kind=orc.KIND_DFA)
                .find_overlapping_iter(sparse, as_numpy=True), "sparse")
    rep = np.tile(np.frombuffer(b"".join(pats[:64]), dtype=np.uint8), 4000)
    want_rep = o.find_overlapping_iter(rep, as_numpy=True)
    assert len(want_rep) > 200000
    for call in range(2):
        assert_same(b.find_overlapping_iter(dev(rep), as_numpy=True), want_rep, f"dense after sparse, call {call}")


def test_event_order_clustered_occurrences():
    """Few occurrences for the span (the order pass takes large buckets: about four events each on average) but all of them
    in a few clusters: buckets of thousands of events, ordered by the second level of k_eo_emit_large with bins of more than
    one end position.  find_iter over the same stream takes the selection without scan launches."""
    import itertools
    pats = [bytes(t) for k in (2, 3) for t in itertools.product(b"abcd", repeat=k)] + orc.gen_patterns(200, seed=0xE0, lo=0x61, span=4)
    pats = list(dict.fromkeys(pats))
    n = 512 << 20                                                        # -> buckets of 64 KiB, bins of 32 end positions
    hay = orc.gen_haystack(0, n, seed=0xE1, lo=0x30, span=10).copy()     # digits: no occurrence outside the clusters
    rng = np.random.default_rng(0xE2)
    for at in (12345, (5 << 20) + 1000, (5 << 20) + 9000, (331 << 20) + 65500, n - 3000):
        hay[at:at + 2600] = rng.integers(0x61, 0x65, 2600, dtype=np.uint8)
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert 16384 < len(want) < n // 1024
    d = dev(hay)
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_engine("pf").build(pats)
    for call in range(2):
        assert_same(a.find_overlapping_iter(d, as_numpy=True), want, f"call {call}")
    out = torch.zeros(len(want) * 24, dtype=torch.uint8, device="cuda")
    m, ok = a.overlapping_device(d, out=out)
    assert ok and m == len(want)
    assert_same(out.cpu().numpy().view(ac.MATCH_DTYPE), want, "device output")
    for mk in (1, 2, 0):
        lf = ac.AhoCorasick.builder().match_kind(mk).kind(ac.AhoCorasickKind.DFA).build(pats)
        olf = orc.Oracle(pats, match_kind=mk, kind=orc.KIND_DFA)
        assert_same(lf.find_iter(d, as_numpy=True), olf.find_iter(hay, as_numpy=True), f"find_iter kind {mk}")


@pytest.mark.parametrize("kind", ["dfa", None])
def test_start_kind_both_unanchored_side_runs_the_filter_engine(c2_patterns, kind):
    """StartKind::Both automata: unanchored searches go through the twin automaton with an unanchored start (same
    noncontiguous NFA => same stream), so they get the prefix filter instead of the walk of the interleaved two-start
    DFA (src/dfa.rs:617-724); anchored searches keep the automaton's own tables."""
    n = 1 << 22
    hay = orc.gen_haystack(0, n, seed=0xAC02)
    plant(hay, c2_patterns[:40], [0, 5, 1000] + [65521 * k for k in range(1, 60)])
    a, o = build_pair(c2_patterns, "standard", {"kind": kind, "start_kind": "both"})
    prof = ac._lib.CProfile()
    got = a.find_overlapping_iter(dev(hay), as_numpy=True, profile=prof)
    assert_same(got, o.find_overlapping_iter(hay, as_numpy=True), "both/unanchored overlapping")
    assert int(prof.engine_used) == 4
    assert_same(a.find_iter(dev(hay), as_numpy=True), o.find_iter(hay, as_numpy=True), "both/unanchored find_iter")
    anch = ac.Input(dev(hay)).anchored(ac.Anchored.Yes)
    assert_same(a.find_iter(anch, as_numpy=True), o.find_iter(hay, anchored=True, as_numpy=True), "both/anchored find_iter")
    w = o.find(hay, anchored=True)
    g = a.find(anch)
    assert (g is None) == (w is None) and (g is None or (g.pattern(), g.start(), g.end()) == tuple(w))
