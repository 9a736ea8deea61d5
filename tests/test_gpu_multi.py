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
