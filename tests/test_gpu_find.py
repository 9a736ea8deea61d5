"""-m gpu: one-shot searches (acgpu_find / acgpu_is_match) through the parallel windowed path vs the oracle's
try_find_fwd (src/automaton.rs:1259-1420): first matches far into large haystacks, at window seams, and where the
leftmost rule needs bytes beyond the window."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
from gpu_util import assert_same, build_pair
from oracle import orc

pytestmark = pytest.mark.gpu
W0 = 16 << 20   # first window of find_parallel


def dev(h):
    return torch.from_numpy(h).cuda()


def same(a, o, hay, dh, span=None):
    inp = ac.Input(dh)
    if span is not None:
        inp.range(*span)
    got = a.find(inp)
    want = o.find(hay, span=span)
    assert (got is None) == (want is None), (got, want, span)
    if got is not None:
        assert got.as_tuple() == tuple(want), (got, want, span)
    assert a.is_match(inp) == (want is not None)


@pytest.mark.parametrize("mk", ["standard", "leftmost_first", "leftmost_longest"])
def test_find_far_and_at_window_seams(mk):
    n = 40 << 20
    hay = orc.gen_haystack(0, n, seed=0xAC09, lo=0x30, span=10)          # digits: no letter pattern occurs
    pats = [b"needle", b"needles", b"need", b"eedlesx", b"dle", b"xyzzyxyzzyxyzzy"]
    a, o = build_pair(pats, mk)
    dh = torch.from_numpy(hay).cuda()
    same(a, o, hay, dh)                                                    # no match at all: scans everything
    for pos in (W0 - 3, W0 - 7, W0, W0 + 1, 2 * W0 + 11, n - 7, 123):  # straddling / just past the first seam, far, end
        h2 = hay.copy()
        h2[pos:pos + 7] = np.frombuffer(b"needles", dtype=np.uint8)
        d2 = torch.from_numpy(h2).cuda()
        same(a, o, h2, d2)
        same(a, o, h2, d2, span=(pos - 50, min(n, pos + 5)))               # span cuts the longer alternatives
        same(a, o, h2, d2, span=(pos + 1, n))                              # starts inside the occurrence
    # the leftmost choice needs bytes beyond the first window: "need" ends inside, "needles" ends outside
    h3 = hay.copy()
    h3[W0 - 4:W0 + 3] = np.frombuffer(b"needles", dtype=np.uint8)
    same(a, o, h3, torch.from_numpy(h3).cuda())


def test_find_dense_small():
    rng = np.random.default_rng(17)
    for case in range(60):
        sigma = int(rng.integers(2, 5))
        pats = [bytes(rng.integers(0x61, 0x61 + sigma, size=int(rng.integers(1, 6)), dtype=np.uint8))
                for _ in range(int(rng.integers(1, 9)))]
        mk = ["standard", "leftmost_first", "leftmost_longest"][case % 3]
        a, o = build_pair(pats, mk, {"kind": [None, "dfa", "cnfa", "nnfa"][case % 4]})
        hay = rng.integers(0x61, 0x61 + sigma + 2, size=int(rng.integers(0, 400)), dtype=np.uint8)
        dh = torch.from_numpy(hay).cuda() if len(hay) else torch.zeros(0, dtype=torch.uint8, device="cuda")
        same(a, o, hay, dh)
        if len(hay) > 4:
            s = int(rng.integers(0, len(hay)))
            e = int(rng.integers(s, len(hay) + 1))
            same(a, o, hay, dh, span=(s, e))


@pytest.mark.parametrize("mk", ["standard", "leftmost_first", "leftmost_longest"])
def test_find_iter_in_windows_when_the_occurrence_stream_does_not_fit(mk):
    """The windowed form of the parallel find_iter (taken when the occurrence stream of the whole span exhausts device
    memory; forced here): matches chosen per window are final, seams carry the end of the last match."""
    rng = np.random.default_rng(11)
    pats = [bytes(rng.integers(0x61, 0x64, size=int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(400)]
    pats += [b"abcabcabcabcabcabcabcabcabcabcabcabcabcabc", b"cccccccccccccccccccccccccc"]
    a, o = build_pair(pats, mk, {"kind": "dfa"}, variants={"find_iter_windows": 1})
    for n, seed in ((0, 1), (5, 2), (70000, 3), (1 << 20, 4)):
        hay = np.random.default_rng(seed).integers(0x61, 0x65, size=n, dtype=np.uint8)   # 'd' never matches: gaps
        if n > 1000:
            hay[n // 2: n // 2 + 42] = np.frombuffer(pats[-2], dtype=np.uint8)
        want = o.find_iter(hay, as_numpy=True)
        got = a.find_iter(dev(hay) if n else hay, as_numpy=True)
        got_span = a.find_iter(ac.Input(dev(hay)).range(n // 3, n - n // 5), as_numpy=True) if n > 10 else None
        assert_same(got, want, f"{mk} n={n} windows")
        if got_span is not None:
            assert_same(got_span, o.find_iter(hay, span=(n // 3, n - n // 5), as_numpy=True), f"{mk} n={n} span windows")
    # sparse matches: most windows select nothing, the floor still advances
    pats2 = [b"needle", b"needles", b"dle"]
    a2, o2 = build_pair(pats2, mk, {"kind": "dfa"}, variants={"find_iter_windows": 1})
    hay = np.full(3 << 20, 0x2E, dtype=np.uint8)
    for at in (0, 4090, 4093, 1 << 20, (3 << 20) - 7):
        hay[at:at + 7] = np.frombuffer(b"needles", dtype=np.uint8)
    got = a2.find_iter(dev(hay), as_numpy=True)
    assert_same(got, o2.find_iter(hay, as_numpy=True), f"{mk} sparse windows")


@pytest.mark.parametrize("mk", ["standard", "leftmost_first"])
def test_find_iter_with_tens_of_occurrences_per_byte_never_materialises_them(mk):
    """40 duplicate patterns per letter: 40 occurrences per byte (42 M for 1 MiB).  Materialising that stream to select
    one match per byte from it would cost far more than the alternatives: a leftmost automaton selects from the per-start
    candidate table (start_select.hip, reported as the prefix filter's trie: engine 4), a Standard one runs the reference
    loop on one lane."""
    pats = [b"a"] * 40 + [b"b"] * 40 + [b"ab", b"ba"]
    a, o = build_pair(pats, mk, {"kind": "dfa"})
    hay = np.random.default_rng(3).integers(0x61, 0x63, size=1 << 20, dtype=np.uint8)
    prof = ac._lib.CProfile()
    got = a.find_iter(dev(hay), as_numpy=True, profile=prof)
    assert_same(got, o.find_iter(hay, as_numpy=True), f"{mk} dense")
    assert int(prof.engine_used) in ((1, 2) if mk == "standard" else (4,))
    same(a, o, hay, dev(hay))
    same(a, o, hay, dev(hay), span=(12345, 99999))


@pytest.mark.parametrize("mk", ["standard", "leftmost_first", "leftmost_longest"])
@pytest.mark.parametrize("kind", ["dfa", "cnfa", None])
def test_find_iter_with_input_earliest(mk, kind):
    """find_iter(Input(h).earliest(true)): the iterator keeps the caller's Input (src/automaton.rs:864-883), so on a
    leftmost automaton every step stops at the first match state entered (:1266) -- e.g. [abcd, b] on "abcd" gives b@1..2,
    not abcd: the Standard rule over the same patterns (capi_find.cpp; tests/test_oracle_naive.py checks the equivalence on
    the oracle), served by the occurrence stream and the selection like every other find_iter."""
    a, o = build_pair([b"abcd", b"b", b"cd"], mk, {"kind": kind})
    h = np.frombuffer(b"abcdabxcd" * 50, dtype=np.uint8).copy()
    for dh in (h, dev(h)):
        got = a.find_iter(ac.Input(dh).earliest(True), as_numpy=True)
        assert_same(got, o.find_iter(h, earliest=True, as_numpy=True), f"earliest {mk} {kind}")
        assert_same(a.find_iter(ac.Input(dh), as_numpy=True), o.find_iter(h, as_numpy=True), f"plain {mk} {kind}")
    rng = np.random.default_rng(23)
    pats = [bytes(rng.integers(0x61, 0x64, size=int(rng.integers(1, 6)), dtype=np.uint8)) for _ in range(12)]
    hay = rng.integers(0x61, 0x64, size=20000, dtype=np.uint8)
    a, o = build_pair(pats, mk, {"kind": kind})
    assert_same(a.find_iter(ac.Input(dev(hay)).earliest(True), as_numpy=True), o.find_iter(hay, earliest=True, as_numpy=True),
                f"random earliest {mk} {kind}")
    assert_same(a.find_iter(ac.Input(dev(hay)).earliest(True).range(17, 15001), as_numpy=True),
                o.find_iter(hay, span=(17, 15001), earliest=True, as_numpy=True), f"random earliest span {mk} {kind}")


@pytest.mark.parametrize("mk", ["leftmost_first", "leftmost_longest"])
def test_earliest_on_a_leftmost_automaton_runs_in_parallel(mk):
    """find / find_iter / is_match with Input::earliest over 1 GiB: round 5 walked it on one lane (~30 ns per byte: half a
    minute for a haystack without a match)."""
    import time
    pats = [b"Sherlock Holmes", b"Holmes", b"Watson said", b"said", b"lock"]   # ("lock" ends inside "Sherlock Holmes": the rules differ)
    a, o = build_pair(pats, mk)
    n = 1 << 30
    hay = torch.full((n,), 0x78, dtype=torch.uint8, device="cuda")   # no match anywhere
    inp = ac.Input(hay).earliest(True)
    assert a.find(inp) is None
    assert len(a.find_iter(inp, as_numpy=True)) == 0   # (untimed: the first call of a size allocates its scratch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert a.find(inp) is None
    dt_find = time.perf_counter() - t0
    t0 = time.perf_counter()
    assert len(a.find_iter(inp, as_numpy=True)) == 0
    dt_iter = time.perf_counter() - t0
    assert dt_find < 0.005 and dt_iter < 0.005, (dt_find, dt_iter)
    # ... and with matches, against the oracle's reference loop (a sample of prose with the names in it, tiled)
    import corpora
    text = corpora.haystack("sherlock.txt", 8 << 20)
    want = o.find_iter(text, earliest=True, as_numpy=True)
    d = dev(text)
    assert_same(a.find_iter(ac.Input(d).earliest(True), as_numpy=True), want, f"earliest {mk} prose")
    m = a.find(ac.Input(d).earliest(True))
    assert (m.pattern(), m.start(), m.end()) == (int(want["pattern"][0]), int(want["start"][0]), int(want["end"][0]))
    plain = o.find_iter(text, as_numpy=True)
    assert len(plain) != len(want) or not all(np.array_equal(plain[f], want[f]) for f in ("pattern", "start", "end"))   # (the two rules differ on this input)


@pytest.mark.parametrize("mk", ["leftmost_first", "leftmost_longest", "standard"])
@pytest.mark.parametrize("deterministic", [False, True])
def test_find_iter_sized_by_the_last_stream(mk, deterministic):
    """Repeated find_iter calls of one automaton: from the second call on scan, event order and selection are queued sized by
    twice the last occurrence stream and synchronised once (capi_find.cpp: nonoverlapping_guessed) -- over a stream above and
    one below the all-pairs limit, then a haystack with four times the occurrences (the guess does not hold: the regular path
    repeats the search), the sparse one again, a sub-span, host and device output.  With deterministic_routing nothing is
    guessed; the results are the same."""
    pats = orc.gen_patterns(400, seed=0xF1, lo=0x61, span=10)
    n = 64 << 20
    sparse = orc.gen_haystack(0, n, seed=0xF2, lo=0x61, span=10)
    few = orc.gen_haystack(0, 2 << 20, seed=0xF3, lo=0x61, span=10)
    from gpu_util import plant
    short = [p for p in pats if len(p) <= 5]
    dense = plant(orc.gen_haystack(0, n, seed=0xF4, lo=0x61, span=10).copy(), short, range(7, n - 16, 64))   # + one occurrence per 64 bytes
    b = ac.AhoCorasick.builder().match_kind(gpu_util_mk(mk)).kind(ac.AhoCorasickKind.DFA)
    if deterministic:
        b.gpu_deterministic_routing(True)
    a = b.build(pats)
    o = orc.Oracle(pats, match_kind=gpu_util_mk(mk), kind=orc.KIND_DFA)
    hays = {"sparse": sparse, "few": few, "dense": dense}
    want = {k: o.find_iter(h, as_numpy=True) for k, h in hays.items()}
    assert len(want["sparse"]) > 16384 and len(want["few"]) < 16384 and len(want["dense"]) > 2 * len(want["sparse"]) + 65536
    dv = {k: dev(h) for k, h in hays.items()}
    out = torch.zeros(len(want["dense"]) * 24 + 24, dtype=torch.uint8, device="cuda")
    for step, k in enumerate(["sparse", "sparse", "sparse", "few", "few", "sparse", "dense", "dense", "sparse", "sparse", "few"]):
        assert_same(a.find_iter(dv[k], as_numpy=True), want[k], f"step {step} ({k}), host output")
        m, ok = a.find_iter_device(dv[k], out)
        assert ok and m == len(want[k]), (step, k, m, len(want[k]))
        assert_same(out[:m * 24].cpu().numpy().view(ac.MATCH_DTYPE), want[k], f"step {step} ({k}), device output")
    lo, hi = (n // 5) | 3, (4 * n // 5) | 1
    sub = o.find_iter(sparse, span=(lo, hi), as_numpy=True)
    for _ in range(2):
        assert_same(a.find_iter(ac.Input(dv["sparse"]).range(lo, hi), as_numpy=True), sub, "sub-span")
    small = torch.zeros(1000 * 24, dtype=torch.uint8, device="cuda")   # a buffer the selection does not fit: the count is reported
    m, ok = a.find_iter_device(dv["sparse"], small)
    assert not ok and m == len(want["sparse"])


def gpu_util_mk(mk):
    from gpu_util import MK
    return MK[mk]
