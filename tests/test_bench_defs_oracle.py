"""not-gpu: the reference's own benchmark definitions (benchmarks/definitions/{sherlock,teddy,curated,same,jetscii}.toml,
definitions/random/{many,misc,memchr}.toml -- pattern sets of 1 to 5 000 patterns, many of 1-3 bytes, haystacks in
English, Russian and Chinese; tests/golden/corpora/bench_defs.json) carry an expected match count per engine: golden
numbers the reference holds for exactly these inputs.  The oracle must reproduce every one of them (find_iter under the
engine's MatchKind; the overlapping count for the one `overlapping` engine)."""
import pytest

import corpora
from oracle import orc

DEFS = corpora.bench_defs()
KINDS = {"standard": 0, "leftmost-first": 1, "leftmost-longest": 2}


@pytest.mark.parametrize("family", sorted(DEFS))
def test_oracle_reproduces_the_reference_counts(family):
    checked = 0
    for b in DEFS[family]:
        pats, hay = corpora.bench_patterns(b), corpora.bench_haystack(b)
        for kind_name, mk in KINDS.items():
            # the configurations of the crate itself that run this bench under this MatchKind
            wants = {corpora.bench_expected(b, e) for e in b["engines"]
                     if e.startswith("rust/aho-corasick/") and e.endswith("/" + kind_name) and "/packed/" not in e}
            wants.discard(None)
            if not wants:
                continue
            assert len(wants) == 1, (family, b["name"], wants)
            o = orc.Oracle(pats, match_kind=mk)
            got = len(o.find_iter(hay, as_numpy=True))
            assert got == wants.pop(), (family, b["name"], kind_name, got)
            checked += 1
        want_ov = corpora.bench_expected(b, "rust/aho-corasick/default/overlapping")
        if want_ov is not None:
            assert len(orc.Oracle(pats).find_overlapping_iter(hay, as_numpy=True)) == want_ov, (family, b["name"])
            checked += 1
    assert checked >= 1
