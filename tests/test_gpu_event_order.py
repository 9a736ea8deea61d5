"""-m gpu: the bucket order pass behind the prefix filters' events (csrc/device/event_order.hip) in both of its chains -- the
fused one (histogram inside the scan kernels, one-launch scan of the bucket words, totals reported by its last kernel, bucket
words re-zeroed behind it; taken by device-to-device calls while the automaton's results were dense) and the one of separate
launches (variant eo_fused = 0) -- against the oracle's overlapping stream (/root/reference/src/automaton.rs:1021-1053)."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
import corpora
from gpu_util import assert_same
from oracle import orc

pytestmark = pytest.mark.gpu


def records(out, n):
    return out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE)


def repeated_calls(a, o, hays, label, spans=()):
    """Device-to-device calls over `hays` in turn, three rounds: the first call of an automaton takes the regular path and
    notes the dense result, the following ones the enqueue machinery with the order pass queued behind the scan."""
    wants = [o.find_overlapping_iter(h, as_numpy=True) for h in hays]
    devs = [torch.from_numpy(h).cuda() for h in hays]
    out = torch.empty(max(len(w) for w in wants) * 24 + 24, dtype=torch.uint8, device="cuda")
    for rnd in range(3):
        for k, d in enumerate(devs):
            out.fill_(0xEE)
            m, ok = a.overlapping_device(d, out=out)
            assert ok and m == len(wants[k]), (label, rnd, k, m, len(wants[k]))
            assert_same(records(out, m), wants[k], f"{label}: round {rnd}, haystack {k}")
    # a buffer that is too small: counted, nothing usable written, and the next fitting call is served
    tiny = torch.full((24 * 7,), 0xEE, dtype=torch.uint8, device="cuda")
    m, ok = a.overlapping_device(devs[0], out=tiny)
    assert not ok and m == len(wants[0])
    m, ok = a.overlapping_device(devs[0], out=out)
    assert ok
    assert_same(records(out, m), wants[0], f"{label}: after a too-small buffer")
    for (s0, s1, b0, b1) in spans:   # span with a shard inside it
        w = o.find_overlapping_iter(hays[0], span=(s0, s1), as_numpy=True)
        w = w[(w["end"] > b0) & (w["end"] <= b1)]
        m, ok = a.overlapping_device(devs[0], span=(s0, s1), shard=(b0, b1), out=out)
        assert ok and m == len(w)
        assert_same(records(out, m), w, f"{label}: span ({s0},{s1}) shard ({b0},{b1})")


@pytest.mark.parametrize("fused", [1, 0])
def test_dense_random_text_two_type_filter(fused):
    """1 000 lowercase patterns over lowercase text (hundreds of thousands of occurrences, a few per 2 KiB bucket), haystacks
    of different sizes in turn: the zero region of the order pass grows and shrinks between calls."""
    pats = orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_engine("pf").build(pats)
    a.set_variant("eo_fused", fused)
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    hays = [orc.gen_haystack(0, n, seed=0xAC02 + k, lo=0x61, span=26) for k, n in enumerate((48 << 20, (160 << 20) + 77, 8 << 20))]
    n0 = len(hays[0])
    repeated_calls(a, o, hays, f"fused={fused}", spans=[(1000, n0 - 5, n0 // 3 + 1, 2 * n0 // 3)])


@pytest.mark.parametrize("fused", [1, 0])
def test_saturated_text_large_buckets(fused):
    """Every position ends several occurrences: thousands of events per bucket (the second bucket level of k_eo_emit_large,
    whose last workgroup closes the fused chain), and a haystack with few events in a few clusters."""
    pats = [b"abab", b"ab", b"b", b"abababab", b"ba", b"bab"]
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_engine("pf").build(pats)
    a.set_variant("eo_fused", fused)
    a.set_variant("routing", 0)
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    rng = np.random.default_rng(0xE2)
    mixed = orc.gen_haystack(0, 24 << 20, seed=0xE1, lo=0x30, span=10).copy()      # digits: no occurrence outside the clusters
    for at in (12345, (5 << 20) + 1000, (5 << 20) + 9000, (23 << 20) + 65500, len(mixed) - 3000):
        mixed[at:at + 2600] = rng.integers(0x61, 0x63, 2600, dtype=np.uint8)
    sat = orc.gen_haystack(0, 3 << 20, seed=0xE3, lo=0x30, span=10).copy()
    sat[(1 << 20) + 333:(1 << 20) + 333 + (16 << 10)] = np.frombuffer(b"ab" * (8 << 10), dtype=np.uint8)
    repeated_calls(a, o, [mixed, sat], f"saturated fused={fused}")


@pytest.mark.parametrize("fused", [1, 0])
def test_natural_text_large_set_filter(fused):
    """English prose against a dictionary (the reference's words-5000 over sherlock.txt): the large-set filter's verifier
    wavefronts do the histogram when they flush their events."""
    words = corpora.words("words-5000")
    a = ac.AhoCorasick.builder().gpu_engine("pf").build(words)
    a.set_variant("eo_fused", fused)
    a.set_variant("pfx_min_patterns", 1)
    o = orc.Oracle(words, kind=orc.KIND_DFA)
    text = corpora.haystack("sherlock.txt")
    hays = [np.tile(text, -(-n // len(text)))[:n].copy() for n in (40 << 20, (64 << 20) + 4321)]
    repeated_calls(a, o, hays, f"natural text fused={fused}")


def test_fused_chain_is_taken_and_reports_like_the_regular_path():
    """The enqueue-only entry point while the automaton is remembered as dense: totals = {records, 0} and the records in
    place; a buffer that cannot hold them: {records, UINT64_MAX}, buffer untouched."""
    pats = orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_engine("pf").build(pats)
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    hay = orc.gen_haystack(0, 160 << 20, seed=0xAC05, lo=0x61, span=26)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > ac.AhoCorasick.ENQUEUE_MAX_EVENTS
    d = torch.from_numpy(hay).cuda()
    out = torch.full((len(want) * 24,), 0xEE, dtype=torch.uint8, device="cuda")
    m, ok = a.overlapping_device(d, out=out)     # (the synchronous call remembers the dense result)
    assert ok and m == len(want)
    tot = torch.zeros(2, dtype=torch.int64, device="cuda")
    for _ in range(3):                           # queued back to back: the bucket words are re-zeroed between them on the stream
        out.fill_(0xEE)
        a.overlapping_enqueue(d, out, tot)
    torch.cuda.synchronize()
    assert int(tot[0]) == len(want) and int(tot[1]) == 0
    assert_same(records(out, len(want)), want, "fused chain through the enqueue-only call")
    tiny = torch.full((24 * 100,), 0xEE, dtype=torch.uint8, device="cuda")
    a.overlapping_enqueue(d, tiny, tot)
    torch.cuda.synchronize()
    assert int(tot[0]) == len(want) and int(tot[1]) == -1 and int(tiny.min()) == 0xEE
