"""-m gpu: BASELINE configs 2, 4 and 5 at FULL size against the CPU oracle -- the whole ordered record stream, not a
prefix.  The oracle's chunk-parallel form (orc.find_overlapping_parallel: the reference loop over every host core,
pieces joined by the max_pattern_len-1 seam rule; tests/test_oracle_parallel.py pins it to the sequential loop) makes
8 GiB a matter of seconds; find_iter (config 5) is inherently sequential and runs the oracle's FindIter on one core."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
from gpu_util import assert_same, build_pair
from oracle import orc

pytestmark = pytest.mark.gpu
GIB = 1 << 30


def plant_dev(buf, pats, positions):
    for i, pos in enumerate(positions):
        p = pats[i % len(pats)]
        if 0 <= pos and pos + len(p) <= buf.numel():
            buf[pos:pos + len(p)] = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()


@pytest.fixture(scope="module")
def c2_patterns():
    return orc.gen_patterns(1000, seed=0xAC01)


@pytest.fixture(scope="module")
def hay8(c2_patterns):
    """The headline haystack: 8 GiB of splitmix random ASCII generated on the device (seed 0xAC02) with occurrences
    planted across 2 KiB / 4 KiB lane-chunk seams, 40-row task seams of the prefix filter and the 16 KiB wave regions of
    the LDS walk engine all over it; plus its host copy for the oracle."""
    n = 8 * GIB
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    ac.gen_haystack(buf, offset=0, seed=0xAC02)
    pos = [((j + 1) * n // 49) // 4096 * 4096 - (j % 13) for j in range(48)]
    pos += [(j + 1) * (n // 37) // (40 * 1008) * (40 * 1008) - (j % 11) for j in range(36)]
    pos += [(j + 1) * (n // 29) // 16384 * 16384 - (j % 17) for j in range(28)]
    plant_dev(buf, [c2_patterns[(5 * j) % len(c2_patterns)] for j in range(len(pos))], pos)
    torch.cuda.synchronize()
    host = np.empty(n, dtype=np.uint8)
    step = GIB
    for o in range(0, n, step):   # chunked copy: no 8 GiB pinned staging buffer
        host[o:o + step] = buf[o:o + step].cpu().numpy()
    return buf, host


def test_c2_8gib_whole_stream_vs_oracle(c2_patterns, hay8):
    """BASELINE configs[1] at full size: every engine's ordered stream over all 8 GiB equals the oracle's."""
    buf, host = hay8
    n = buf.numel()
    o = orc.Oracle(c2_patterns, kind=orc.KIND_DFA)
    want, want_hash = o.find_overlapping_parallel(host)
    assert len(want) > 7000
    for engine, chunk in (("pf", 0), ("hot", 0), ("hot", 4096), ("walk", 4096)):
        a, _ = build_pair(c2_patterns, "standard", {"kind": "dfa"}, chunk=chunk, engine=engine)
        got = a.find_overlapping_iter(buf, as_numpy=True)
        assert_same(got, want, f"8 GiB {engine} chunk={chunk} vs oracle")
        assert orc.hash_matches(got) == want_hash
    a, _ = build_pair(c2_patterns, "standard", {"kind": "dfa"})
    mid = (n // 2) - 37
    parts = [a.find_overlapping_shard(ac.Input(buf), 0, mid), a.find_overlapping_shard(ac.Input(buf), mid, n)]
    assert_same(np.concatenate(parts), want, "8 GiB two shards vs oracle")
    # the pipelined (enqueue-only) form bench.py times
    out = torch.empty(len(want) * 24 + 4096, dtype=torch.uint8, device="cuda")
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    a.overlapping_enqueue(buf, out, totals)
    torch.cuda.synchronize()
    t = totals.cpu().numpy()
    assert int(t[0]) == len(want) and int(t[1]) <= a.ENQUEUE_MAX_EVENTS
    assert_same(out[: len(want) * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, "8 GiB enqueue form vs oracle")


def plant_both(buf, host, pats, positions):
    """The same occurrences into the device haystack and its host copy (the fixture stays consistent for later tests)."""
    plant_dev(buf, pats, positions)
    for i, pos in enumerate(positions):
        p = pats[i % len(pats)]
        if 0 <= pos and pos + len(p) <= len(host):
            host[pos:pos + len(p)] = np.frombuffer(p, dtype=np.uint8)
    torch.cuda.synchronize()


def test_c4_100k_patterns_8gib_vs_oracle(hay8):
    """BASELINE configs[3] at full size: 100 000 patterns, AhoCorasickKind::ContiguousNFA, over all 8 GiB of the headline
    haystack with occurrences planted across lane-chunk seams: the contiguous-NFA failure-link walk
    (src/nfa/contiguous.rs:186-247) on the device ("walk": k_cnfa_tri + the records from its match events, and the same
    walk with the re-walking fill) and the default engine each equal the oracle's contiguous-NFA stream (chunk-parallel
    reference loop), record for record."""
    buf, host = hay8
    n = buf.numel()
    pats = orc.gen_patterns(100000, seed=0xAC04)
    pos = [(j + 1) * (n // 521) // 2048 * 2048 - (j % 15) for j in range(520)]
    plant_both(buf, host, pats[::193], pos)
    o = orc.Oracle(pats, kind=orc.KIND_CNFA)
    want, want_hash = o.find_overlapping_parallel(host)
    assert len(want) > 800000
    for engine in ("auto", "walk"):
        a, _ = build_pair(pats, "standard", {"kind": "cnfa"}, engine=engine)
        assert a.kind() == ac.AhoCorasickKind.ContiguousNFA
        got = a.find_overlapping_iter(buf, as_numpy=True)
        assert_same(got, want, f"c4 8 GiB {engine} vs oracle")
        assert orc.hash_matches(got) == want_hash
    # the walk's enqueue-only form (count pass with match events -> scan -> emit, sizes read on the device)
    a, _ = build_pair(pats, "standard", {"kind": "cnfa"}, engine="walk")
    out = torch.empty(len(want) * 24 + 4096, dtype=torch.uint8, device="cuda")
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    a.overlapping_enqueue(buf, out, totals)
    torch.cuda.synchronize()
    assert int(totals.cpu().numpy()[0]) == len(want)
    assert_same(out[: len(want) * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, "c4 8 GiB walk, enqueue form vs oracle")


def test_c5_casei_leftmost_first_8gib_vs_oracle(c2_patterns, hay8):
    """BASELINE configs[4] at full size: 1 000 patterns, ascii_case_insensitive + LeftmostFirst, find_iter over all 8 GiB
    (mixed-case occurrences planted): the parallel selection on the device equals the oracle's FindIter
    (src/automaton.rs:857-936 over try_find_fwd :1285-1420, one core, ~20 s)."""
    buf, host = hay8
    n = buf.numel()
    planted = [p.swapcase() if j % 2 else p.lower() for j, p in enumerate(c2_patterns[:97])]
    pos = [(j + 1) * (n // 1601) - (j % 23) for j in range(1600)]
    plant_both(buf, host, planted, pos)
    a, o = build_pair(c2_patterns, "leftmost_first", {"kind": "dfa", "ascii_case_insensitive": True})
    want = o.find_iter(host, as_numpy=True)
    assert len(want) > 40000
    assert_same(a.find_iter(buf, as_numpy=True), want, "c5 8 GiB vs oracle")
    b, _ = build_pair(c2_patterns, "leftmost_first", {"kind": "dfa", "ascii_case_insensitive": True}, engine="hot")
    assert_same(b.find_iter(buf, as_numpy=True), want, "c5 8 GiB, LDS walk engine vs oracle")


def test_natural_text_beyond_4gib_large_set_filter():
    """The reference's own benchmark inputs at a size that crosses 2^32: English prose (sherlock.txt tiled to 4.5 GiB)
    against words-5000 -- the two-type filter abandons, the large-set filter with its long-prefix level 2 and second-pass
    level 3 takes over (acgpu_profile.routed) -- and the same kernels requested directly; whole stream vs the oracle."""
    import corpora
    text = corpora.haystack("sherlock.txt")
    n = 4 * GIB + GIB // 2 + 12345
    host = np.tile(text, -(-n // len(text)))[:n].copy()
    words = corpora.words("words-5000")
    o = orc.Oracle(words, kind=orc.KIND_DFA)
    want, want_hash = o.find_overlapping_parallel(host)
    assert len(want) > 4_000_000 and int(want["end"][-1]) > 1 << 32
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    for o0 in range(0, n, GIB):
        buf[o0:o0 + GIB] = torch.from_numpy(host[o0:o0 + GIB]).cuda()
    a, _ = build_pair(words, "standard", {})
    prof = ac._lib.CProfile()
    got = a.find_overlapping_iter(buf, as_numpy=True, profile=prof)
    assert int(prof.engine_used) == 4 and int(prof.routed) == 1
    assert_same(got, want, "4.5 GiB natural text, automatic choice")
    assert orc.hash_matches(got) == want_hash
    a2, _ = build_pair(words, "standard", {}, engine="pf", variants={"pfx_min_patterns": 1})
    got = a2.find_overlapping_iter(buf, as_numpy=True)
    assert orc.hash_matches(got) == want_hash and len(got) == len(want)
