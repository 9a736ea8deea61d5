"""-m gpu: the GPU-side DFA fill (device/dfa_fill.hip) produces the CPU builder's table word for
word (which tests/test_tables_parity.py pins to the oracle's restatement of src/dfa.rs:431-835)."""
import time

import numpy as np
import pytest

import aho_corasick_amd as ac
from oracle import orc

pytestmark = pytest.mark.gpu


def trans_of(a):
    t = a.tables()
    n = int(t.dfa_trans_len)
    return np.ctypeslib.as_array(t.dfa_trans, shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint32)


CASES = [
    dict(mk=0, sk=1, casei=False, bc=True),
    dict(mk=0, sk=2, casei=False, bc=True),     # anchored rows
    dict(mk=1, sk=1, casei=True, bc=True),      # leftmost-first + case folding
    dict(mk=2, sk=2, casei=False, bc=False),    # leftmost-longest, anchored, 256-wide rows
    dict(mk=0, sk=1, casei=False, bc=False),
    dict(mk=0, sk=0, casei=False, bc=True),     # StartKind::Both: both row sets from the device, interleaved copies
    dict(mk=1, sk=0, casei=True, bc=False),
]


@pytest.mark.parametrize("c", CASES)
def test_device_fill_equals_host_fill(c):
    rng = np.random.default_rng(3)
    sets = [orc.gen_patterns(1000, seed=0xAC01),
            [bytes(rng.integers(0x61, 0x64, size=int(rng.integers(1, 9)), dtype=np.uint8)) for _ in range(300)],
            [b"a", b"ab", b"abc", b"bc", b"c", b"abcd" * 40],
            []]
    for pats in sets:
        def build(gpu):
            return (ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).match_kind(c["mk"]).start_kind(c["sk"])
                    .ascii_case_insensitive(c["casei"]).byte_classes(c["bc"]).gpu_dfa_fill(gpu).build(pats))
        host, dev = build(False), build(True)
        th, td = trans_of(host), trans_of(dev)
        assert th.shape == td.shape and np.array_equal(th, td), f"{c} npat={len(pats)}"
        o = orc.Oracle(pats, match_kind=c["mk"], start_kind=c["sk"], kind=orc.KIND_DFA,
                       ascii_case_insensitive=c["casei"], byte_classes=c["bc"])
        assert np.array_equal(td, trans_of(o))
        hay = orc.gen_haystack(0, 20000, seed=5, lo=0x61, span=4)
        for anchored in ((False, True) if c["sk"] == 0 else (c["sk"] == 2,)):
            want = o.find_iter(hay, anchored=anchored, as_numpy=True)
            got = dev.find_iter(ac.Input(hay).anchored(ac.Anchored.Yes if anchored else ac.Anchored.No), as_numpy=True)
            assert np.array_equal(got["start"], want["start"]) and np.array_equal(got["pattern"], want["pattern"])


def test_large_dfa_build_time():
    pats = orc.gen_patterns(20000, seed=0xAC04)     # 160k states x 128 classes = 82 MB table
    t0 = time.perf_counter()
    host = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build(pats)
    t1 = time.perf_counter()
    dev = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_dfa_fill(True).build(pats)
    t2 = time.perf_counter()
    assert np.array_equal(trans_of(host), trans_of(dev))
    print(f"\\n20k-pattern full DFA build: host fill {t1 - t0:.3f} s, device fill {t2 - t1:.3f} s (whole acgpu_build)")
