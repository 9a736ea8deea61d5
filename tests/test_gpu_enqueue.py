"""-m gpu: the enqueue-only form of the overlapping search (acgpu_find_overlapping_enqueue): several calls queued
back to back on one stream without a host round trip produce what the synchronous calls produce."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
from gpu_util import assert_same, plant
from oracle import orc

pytestmark = pytest.mark.gpu


def records(out, n):
    return out[: n * 24].cpu().numpy().view(ac.MATCH_DTYPE)


def test_enqueued_calls_equal_synchronous_calls():
    pats = orc.gen_patterns(1000, seed=0xAC01)
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build(pats)
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    hays = []
    for k in range(5):
        h = orc.gen_haystack(k << 22, (1 << 22) + 1000 * k, seed=0xAC02)
        plant(h, pats[k::50], [4093 * j + k for j in range(1, 200)])
        hays.append(h)
    devs = [torch.from_numpy(h).cuda() for h in hays]
    outs = [torch.zeros(1 << 20, dtype=torch.uint8, device="cuda") for _ in hays]
    tots = torch.zeros((len(hays), 2), dtype=torch.int64, device="cuda")
    for k, d in enumerate(devs):   # queued back to back: no synchronisation in between
        a.overlapping_enqueue(d, outs[k], tots[k], slot=k)
    # a shard of a span, with a cold floor in front of it, on the same stream
    sh_out = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    sh_tot = torch.zeros(2, dtype=torch.int64, device="cuda")
    n0 = len(hays[0])
    a.overlapping_enqueue(devs[0], sh_out, sh_tot, span=(100, n0 - 7), shard=(n0 // 3, 2 * n0 // 3))
    torch.cuda.synchronize()
    t = tots.cpu().numpy()
    for k, h in enumerate(hays):
        want = o.find_overlapping_iter(h, as_numpy=True)
        assert t[k, 0] == len(want) and 0 < t[k, 1] <= ac.AhoCorasick.ENQUEUE_MAX_EVENTS
        assert_same(records(outs[k], int(t[k, 0])), want, f"enqueued call {k}")
        assert 0.0 < a.enqueue_kernel_ms(k) < 50.0
    want = o.find_overlapping_iter(hays[0], span=(100, n0 - 7), as_numpy=True)
    want = want[(want["end"] > n0 // 3) & (want["end"] <= 2 * n0 // 3)]
    assert_same(records(sh_out, int(sh_tot[0])), want, "enqueued shard")


def test_enqueue_dense_results_are_delivered_and_small_buffers_are_not_written():
    pats = orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build(pats)
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    dense = orc.gen_haystack(0, 160 << 20, seed=0xAC02, lo=0x61, span=26)      # > 16384 occurrences
    want = o.find_overlapping_iter(dense, as_numpy=True)
    d = torch.from_numpy(dense).cuda()
    out = torch.full((len(want) * 24,), 0xEE, dtype=torch.uint8, device="cuda")
    tot = torch.zeros(2, dtype=torch.int64, device="cuda")
    a.overlapping_enqueue(d, out, tot)
    a.overlapping_enqueue(d, out, tot)          # (twice: the context re-arms itself after an overflow)
    torch.cuda.synchronize()
    # more occurrences than the all-pairs rank orders, and nothing known about this automaton yet: reported, not written
    assert len(want) > ac.AhoCorasick.ENQUEUE_MAX_EVENTS
    assert int(tot[0]) == len(want) and int(tot[1]) > ac.AhoCorasick.ENQUEUE_MAX_EVENTS
    assert int(out.min()) == 0xEE               # the caller repeats with the synchronous form ...
    m, ok = a.overlapping_device(d, out=out)
    assert ok and m == len(want)
    assert_same(records(out, m), want, "synchronous repeat")
    # ... which remembers the dense result: the next enqueue-only calls queue the bucket order pass behind their scan
    out.fill_(0xEE)
    a.overlapping_enqueue(d, out, tot)
    torch.cuda.synchronize()
    assert int(tot[0]) == len(want) and int(tot[1]) == 0
    assert_same(records(out, len(want)), want, "dense result through the enqueue form")
    # records > cap: counted, not written; then the same context serves a fitting call
    small_h = dense[: 8 << 20]
    w2 = o.find_overlapping_iter(small_h, as_numpy=True)
    tiny = torch.full((24 * 10,), 0xEE, dtype=torch.uint8, device="cuda")
    a.overlapping_enqueue(d[: 8 << 20], tiny, tot)
    torch.cuda.synchronize()
    assert int(tot[0]) == len(w2) > 10 and int(tiny.min()) == 0xEE
    a.overlapping_enqueue(d[: 8 << 20], out, tot)
    torch.cuda.synchronize()
    assert_same(records(out, int(tot[0])), w2, "after a too-small buffer")


def test_enqueue_keeps_the_error_order_and_serves_empty_patterns():
    hay = torch.zeros(1 << 12, dtype=torch.uint8, device="cuda")
    out = torch.zeros(1 << 18, dtype=torch.uint8, device="cuda")
    tot = torch.zeros(2, dtype=torch.int64, device="cuda")
    lf = ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).build([b"ab", b"b"])
    with pytest.raises(ac.MatchError):   # automaton.rs:397-423: overlapping needs MatchKind::Standard
        lf.overlapping_enqueue(hay, out, tot)
    # an automaton with an empty pattern is not the filter's: the enqueue form runs the walk pipeline for it
    pats = [b"", b"ab", b"b"]
    h = np.frombuffer(b"abbaababbab" * 300, dtype=np.uint8).copy()
    empty = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build(pats)
    want = orc.Oracle(pats, kind=orc.KIND_DFA).find_overlapping_iter(h, as_numpy=True)
    empty.overlapping_enqueue(torch.from_numpy(h).cuda(), out, tot)
    torch.cuda.synchronize()
    assert int(tot[0]) == len(want) and int(tot[1]) == 0
    assert_same(records(out, len(want)), want, "empty pattern, enqueue form")


@pytest.mark.parametrize("case", ["dense_classic", "hot", "walk", "cnfa_walk", "large_set"])
def test_enqueue_form_for_every_engine_and_dense_results(case):
    """The enqueue-only form beyond the event path: ACGPU_ENQUEUE_CLASSIC for dense results (no 16 384-occurrence limit),
    and automata served by the LDS walk / global walks / the large-set filter -- all without a host round trip."""
    import aho_corasick_amd as ac
    from gpu_util import build_pair, plant
    n = 4 << 20
    if case == "dense_classic":
        pats = [p[:3] for p in orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)]   # ~230 k occurrences in 4 MiB
        hay = orc.gen_haystack(0, n, seed=0xAC02, lo=0x61, span=26)
        a, o = build_pair(pats, "standard", {"kind": "dfa"})
    else:
        npat = 30000 if case == "large_set" else 1000
        pats = orc.gen_patterns(npat, seed=0xAC01)
        hay = orc.gen_haystack(0, n, seed=0xAC02)
        plant(hay, pats[:64], [4099 * k for k in range(1, 900)])
        kw = {"kind": "cnfa"} if case in ("cnfa_walk", "large_set") else {"kind": "dfa"}
        a, o = build_pair(pats, "standard", kw, engine={"hot": "hot", "walk": "walk", "cnfa_walk": "walk", "large_set": "auto"}[case])
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > (20000 if case == "dense_classic" else 500)
    d = torch.from_numpy(hay).cuda()
    out = torch.zeros(len(want) * 24 + 4096, dtype=torch.uint8, device="cuda")
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    for rep in range(2):   # the second call reuses the per-stream context
        a.overlapping_enqueue(d, out, totals, classic=(case == "dense_classic"))
        torch.cuda.synchronize()
        t = totals.cpu().numpy()
        assert int(t[0]) == len(want) and int(t[1]) <= a.ENQUEUE_MAX_EVENTS, (case, t)
        assert_same(out[: len(want) * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, f"enqueue {case} rep {rep}")
    # a buffer that is too small: nothing usable written, the count says so
    small = torch.zeros(2400, dtype=torch.uint8, device="cuda")
    a.overlapping_enqueue(d, small, totals, classic=(case == "dense_classic"))
    torch.cuda.synchronize()
    assert int(totals.cpu().numpy()[0]) == len(want)
