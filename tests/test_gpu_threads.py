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
