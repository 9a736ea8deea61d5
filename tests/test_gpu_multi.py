"""-m gpu: acgpu_find_overlapping_multi through the Python binding -- virtual shards on cuda:0 against the oracle:
sparse results (every shard finishes in the enqueue-only form), dense results (shards repeated synchronously, growing
buffers), automata the enqueue form does not serve, empty shards."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
from gpu_util import assert_same, build_pair, plant
from oracle import orc

pytestmark = pytest.mark.gpu


def shards_of(hay, cuts, halo):
    t = torch.from_numpy(hay).cuda()
    out = []
    for i, (b, e) in enumerate(zip(cuts[:-1], cuts[1:])):
        left = halo if i else 0
        out.append(t[b - left:e].clone())
    return out


@pytest.mark.parametrize("case", ["sparse", "dense", "cnfa", "empty_middle"])
def test_multi_equals_oracle(case):
    n = 8 << 20
    if case == "dense":
        pats = orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)
        hay = orc.gen_haystack(0, n, seed=0xAC02, lo=0x61, span=26)
        kw = {"kind": "dfa"}
    else:
        pats = orc.gen_patterns(1000, seed=0xAC01)
        hay = orc.gen_haystack(0, n, seed=0xAC02)
        kw = {"kind": "cnfa"} if case == "cnfa" else {"kind": "dfa"}
    cuts = [0, n // 5 + 3, n // 2, n // 2 + 40000, n] if case != "empty_middle" else [0, n // 3, n // 3, n // 3, n]
    plant(hay, pats[:64], [c - d for c in cuts[1:-1] for d in (1, 5, 15)] + [1000 * k for k in range(1, 50)])
    engine = "walk" if case == "cnfa" else "auto"
    a, o = build_pair(pats, "standard", kw, engine=engine)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    halo = a.max_pattern_len() - 1
    sh = shards_of(hay, cuts, halo)   # (empty_middle: two zero-length shards whose buffers are just their halo)
    out = torch.zeros(max(len(want), 1) * 24 + 4096, dtype=torch.uint8, device="cuda")
    m, counts = a.find_overlapping_multi(sh, out)
    assert m == len(want) == sum(counts)
    assert_same(out[: m * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, case)
    if len(want) > 100:
        small = torch.zeros(2400, dtype=torch.uint8, device="cuda")
        with pytest.raises(ValueError):
            a.find_overlapping_multi(sh, small)


def test_multi_rejects_what_it_cannot_shard_and_reports_through_last_error():
    """An automaton with an empty pattern matches at every position, the seam included: with an empty halo a shard cannot
    tell whose match that is, so the sharded search refuses it (the message reaches acgpu_last_error, which the bindings
    read); shards shorter than a halo are refused by the binding."""
    hay = np.frombuffer(b"abcabcabc" * 1000, dtype=np.uint8).copy()
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build([b"", b"a"])
    t = torch.from_numpy(hay).cuda()
    out = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception) as e:
        a.find_overlapping_multi([t[:4000].clone(), t[4000:].clone()], out)
    assert "empty pattern" in str(e.value)
    m, _ = a.find_overlapping_multi([t], out)          # one shard is the plain search
    assert m == len(orc.Oracle([b"", b"a"], kind=orc.KIND_DFA).find_overlapping_iter(hay, as_numpy=True))
    b = ac.AhoCorasick.builder().build([b"abcabcabcabc"])
    with pytest.raises(ValueError):
        b.find_overlapping_multi([t[:3].clone(), t[:100].clone()], out)


def test_multi_from_two_host_threads_on_one_automaton():
    """Two host threads searching the same automaton at once: each call checks its streams out of the pool, so the
    enqueue-only scratch (kept per automaton and stream) is never shared."""
    import threading
    pats = orc.gen_patterns(1000, seed=0xAC01)
    a, o = build_pair(pats, "standard", {"kind": "dfa"})
    halo = a.max_pattern_len() - 1
    results = {}

    def run(k):
        n = (6 + k) << 20
        hay = orc.gen_haystack(k * 977, n, seed=0xAC02 + k)
        plant(hay, pats[:32], [n // 2 - d for d in (1, 7, 15)] + [4096 * j for j in range(1, 40)])
        sh = shards_of(hay, [0, n // 2, n], halo)
        out = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
        for _ in range(6):
            m, _ = a.find_overlapping_multi(sh, out)
            got = out[: m * 24].cpu().numpy().view(ac.MATCH_DTYPE).copy()
            results.setdefault(k, []).append((got, o.find_overlapping_iter(hay, as_numpy=True)))

    ts = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert sorted(results) == [0, 1]
    for k, pairs in results.items():
        for got, want in pairs:
            assert_same(got, want, f"thread {k}")


def test_multi_rccl_between_two_devices():
    """The RCCL transport with two communicator ranks: runs wherever the box exposes two devices (the driver's 1-GPU boxes
    skip it; the first multi-GPU node executes it)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices")
    pats = orc.gen_patterns(1000, seed=0xAC01)
    n = 16 << 20
    hay = orc.gen_haystack(0, n, seed=0xAC02)
    plant(hay, pats[:64], [n // 2 - d for d in (1, 5, 15)] + [1000 * k for k in range(1, 50)])
    a, o = build_pair(pats, "standard", {"kind": "dfa"})
    halo = a.max_pattern_len() - 1
    t = torch.from_numpy(hay)
    sh = [t[: n // 2].to("cuda:0"), t[n // 2 - halo:].to("cuda:1")]
    out = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0")
    m, counts = a.find_overlapping_multi(sh, out)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert m == len(want) == sum(counts)
    assert_same(out[: m * 24].cpu().numpy().view(ac.MATCH_DTYPE), want, "two devices")
    assert ac.load_library().acgpu_multi_last_transport() == 2   # RCCL send/recv
