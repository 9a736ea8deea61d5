"""-m gpu: the reference's own benchmark definitions (benchmarks/definitions/{sherlock,teddy,curated,same,jetscii}.toml,
definitions/random/{many,misc,memchr}.toml; tests/golden/corpora/bench_defs.json) on the device: every pattern set (1 to
5 000 patterns, many of 1-3 bytes, bytes >= 0x80) over its own haystack (English, Russian, Chinese, random, repeated
single bytes) -- the overlapping stream with the automatic engine choice and with the walk engines, and find_iter under
every MatchKind the definition runs -- against the oracle, record for record, and against the reference's expected
count."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
import corpora
from gpu_util import assert_same
from oracle import orc

pytestmark = pytest.mark.gpu
DEFS = corpora.bench_defs()
KINDS = {"standard": (0, ac.MatchKind.Standard), "leftmost-first": (1, ac.MatchKind.LeftmostFirst),
         "leftmost-longest": (2, ac.MatchKind.LeftmostLongest)}


@pytest.mark.parametrize("family", sorted(DEFS))
def test_reference_benchmark_definitions(family):
    for b in DEFS[family]:
        pats, hay = corpora.bench_patterns(b), corpora.bench_haystack(b)
        d = torch.from_numpy(hay).cuda()
        want_ov = orc.Oracle(pats).find_overlapping_iter(hay, as_numpy=True)
        for engine in ("auto", "walk"):
            a = ac.AhoCorasick.builder().gpu_engine(engine).build(pats)
            assert_same(a.find_overlapping_iter(d, as_numpy=True), want_ov, f"{family}/{b['name']} overlapping, engine {engine}")
        ref_ov = corpora.bench_expected(b, "rust/aho-corasick/default/overlapping")
        assert ref_ov is None or ref_ov == len(want_ov)
        for kind_name, (mk, kind) in KINDS.items():
            wants = {corpora.bench_expected(b, e) for e in b["engines"]
                     if e.startswith("rust/aho-corasick/") and e.endswith("/" + kind_name) and "/packed/" not in e}
            wants.discard(None)
            if not wants:
                continue
            want = orc.Oracle(pats, match_kind=mk).find_iter(hay, as_numpy=True)
            got = ac.AhoCorasick.builder().match_kind(kind).build(pats).find_iter(d, as_numpy=True)
            assert_same(got, want, f"{family}/{b['name']} find_iter {kind_name}")
            assert wants == {len(got)}, (family, b["name"], kind_name, wants, len(got))
