"""-m gpu: one automaton shared by concurrent host threads (the reference's `AhoCorasick: Send + Sync`,
src/lib.rs:283-301; SURVEY.md 8b "Threading"): every call leases its own scratch, results stay bit-exact."""
import threading

import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
from gpu_util import assert_same
from oracle import orc

pytestmark = pytest.mark.gpu


def test_concurrent_searches_share_one_automaton():
    pats = orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)
    std = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build(pats)
    lf = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).match_kind(ac.MatchKind.LeftmostFirst).build(pats)
    o_std = orc.Oracle(pats, kind=orc.KIND_DFA)
    o_lf = orc.Oracle(pats, match_kind=1, kind=orc.KIND_DFA)
    repl = [b"<%d>" % (i % 7) for i in range(len(pats))]
    # haystacks of very different match densities and sizes, so the threads sit in different result modes
    # (all-pairs rank, sorted events, classic pipeline after a density jump) at the same time
    hays = [orc.gen_haystack(0, (1 << 20) * (1 + 3 * t), seed=100 + t, lo=0x61, span=26 if t % 2 else 4) for t in range(6)]
    want_ov = [o_std.find_overlapping_iter(h, as_numpy=True) for h in hays]
    want_it = [o_lf.find_iter(h, as_numpy=True) for h in hays]
    want_rp = [orc.replace_all_bytes(o_lf, h, repl) for h in hays[:2]]
    devs = [torch.from_numpy(h).cuda() for h in hays]
    std.upload(0)
    lf.upload(0)
    errors = []

    def worker(t):
        try:
            torch.cuda.set_device(0)
            for rnd in range(4):
                k = (t + rnd) % len(hays)
                assert_same(std.find_overlapping_iter(devs[k], as_numpy=True), want_ov[k], f"thread {t} overlapping {k}")
                assert_same(lf.find_iter(devs[k], as_numpy=True), want_it[k], f"thread {t} find_iter {k}")
                assert_same(std.find_overlapping_iter(hays[k], as_numpy=True), want_ov[k], f"thread {t} host haystack {k}")
                if k < 2:
                    got = bytes(lf.replace_all_bytes(devs[k], repl).cpu().numpy())
                    assert got == want_rp[k], f"thread {t} replace_all {k}"
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]


@pytest.mark.parametrize("deterministic", [False, True])
def test_two_threads_alternate_dense_and_sparse_inputs(deterministic):
    """Two threads issue dense and sparse searches alternately against ONE automaton (overlapping + leftmost find_iter):
    whatever the adaptive hints of the shared DeviceState say at any moment (route / dense / start-table / walk hints --
    or nothing at all with acgpu_config.deterministic_routing), every result is the oracle's."""
    pats = orc.gen_patterns(300, seed=0xAC31, lo=0x61, span=26) + [b"ab", b"b", b"abc"]
    b_std = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_deterministic_routing(deterministic)
    b_lf = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).match_kind(ac.MatchKind.LeftmostFirst) \
        .gpu_deterministic_routing(deterministic)
    std, lf = b_std.build(pats), b_lf.build(pats)
    o_std, o_lf = orc.Oracle(pats, kind=orc.KIND_DFA), orc.Oracle(pats, match_kind=1, kind=orc.KIND_DFA)
    dense = orc.gen_haystack(0, 3 << 20, seed=7, lo=0x61, span=3)          # "ab", "b", "abc" everywhere
    sparse = orc.gen_haystack(0, 20 << 20, seed=8, lo=0x30, span=40)       # digits and capitals: next to nothing
    plant_at = np.arange(5000, len(sparse) - 64, 1 << 18)
    for i, p in enumerate(plant_at):
        w = np.frombuffer(pats[i % len(pats)], dtype=np.uint8)
        sparse[p:p + len(w)] = w
    hays = [dense, sparse]
    want_ov = [o_std.find_overlapping_iter(h, as_numpy=True) for h in hays]
    want_it = [o_lf.find_iter(h, as_numpy=True) for h in hays]
    assert len(want_ov[0]) > 1_000_000 and 50 < len(want_ov[1]) < 5000
    devs = [torch.from_numpy(h).cuda() for h in hays]
    std.upload(0)
    lf.upload(0)
    errors = []

    def worker(t):
        try:
            torch.cuda.set_device(0)
            for rnd in range(10):
                k = (t + rnd) % 2
                assert_same(std.find_overlapping_iter(devs[k], as_numpy=True), want_ov[k], f"thread {t} round {rnd} overlapping {k}")
                assert_same(lf.find_iter(devs[k], as_numpy=True), want_it[k], f"thread {t} round {rnd} find_iter {k}")
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]
