"""The reference's builder-config matrix (src/tests.rs:723-1323), as data.

Each entry: (test-id, collection-or-groups, match_kind, api, builder kwargs).
`api` is one of "find_iter", "overlapping", "anchored" (find_iter with
Input::anchored(Anchored::Yes), src/tests.rs:654-669).
Kwargs use the names of AhoCorasickBuilder's setters (src/ahocorasick.rs:2342-2616).
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
VECTORS = json.load(open(os.path.join(HERE, "golden", "reference_vectors.json")))

DENSE_MAX = 0xFFFFFFFF  # usize::MAX in the reference

# testcombo! src/tests.rs:723-863
COMBO = [
    ("default", {}),
    ("nfa_default", {"kind": "nnfa"}),
    ("nfa_noncontig_no_prefilter", {"kind": "nnfa", "prefilter": False}),
    ("nfa_noncontig_all_sparse", {"kind": "nnfa", "dense_depth": 0}),
    ("nfa_noncontig_all_dense", {"kind": "nnfa", "dense_depth": DENSE_MAX}),
    ("nfa_contig_default", {"kind": "cnfa"}),
    ("nfa_contig_no_prefilter", {"kind": "cnfa", "prefilter": False}),
    ("nfa_contig_all_sparse", {"kind": "cnfa", "dense_depth": 0}),
    ("nfa_contig_all_dense", {"kind": "cnfa", "dense_depth": DENSE_MAX}),
    ("nfa_contig_no_byte_class", {"kind": "cnfa", "byte_classes": False}),
    ("dfa_default", {"kind": "dfa"}),
    ("dfa_start_both", {"kind": "dfa", "start_kind": "both"}),
    ("dfa_no_prefilter", {"kind": "dfa", "prefilter": False}),
    ("dfa_start_both_no_prefilter", {"kind": "dfa", "start_kind": "both", "prefilter": False}),
    ("dfa_no_byte_class", {"kind": "dfa", "byte_classes": False}),
    ("dfa_start_both_no_byte_class", {"kind": "dfa", "start_kind": "both", "byte_classes": False}),
]

# src/tests.rs:876-994
OVERLAPPING_CONFIGS = [
    ("default", {}),
    ("nfa_noncontig_default", {"kind": "nnfa"}),
    ("nfa_noncontig_no_prefilter", {"kind": "nnfa", "prefilter": False}),
    ("nfa_contig_default", {"kind": "cnfa"}),
    ("nfa_contig_no_prefilter", {"kind": "cnfa", "prefilter": False}),
    ("nfa_contig_all_sparse", {"kind": "cnfa", "dense_depth": 0}),
    ("nfa_contig_all_dense", {"kind": "cnfa", "dense_depth": DENSE_MAX}),
    ("dfa_default", {"kind": "dfa"}),
    ("dfa_start_both", {"kind": "dfa", "start_kind": "both"}),
    ("dfa_no_prefilter", {"kind": "dfa", "prefilter": False}),
    ("dfa_start_both_no_prefilter", {"kind": "dfa", "start_kind": "both", "prefilter": False}),
    ("dfa_no_byte_class", {"kind": "dfa", "byte_classes": False}),
    ("dfa_start_both_no_byte_class", {"kind": "dfa", "start_kind": "both", "byte_classes": False}),
]

# src/tests.rs:1039-1179
ANCHORED_CONFIGS = [
    ("default", {"start_kind": "anchored"}),
    ("nfa_noncontig_default", {"start_kind": "anchored", "kind": "nnfa"}),
    ("nfa_contig_default", {"start_kind": "anchored", "kind": "cnfa"}),
    ("dfa_default", {"start_kind": "anchored", "kind": "dfa"}),
    ("dfa_start_both", {"start_kind": "both", "kind": "dfa"}),
]


def _groups(names):
    out = []
    for g in names:
        out.extend(VECTORS["groups"][g])
    return out


def collection(name):
    return _groups(VECTORS["collections"][name])


def all_cases():
    """Yield (id, match_kind, api, kwargs, vectors)."""
    # src/tests.rs:867-873
    for mk, coll in (("leftmost_longest", "AC_LEFTMOST_LONGEST"), ("leftmost_first", "AC_LEFTMOST_FIRST"),
                     ("standard", "AC_STANDARD_NON_OVERLAPPING")):
        for cid, kw in COMBO:
            yield f"search_{mk}::{cid}", mk, "find_iter", kw, collection(coll)
    for cid, kw in OVERLAPPING_CONFIGS:
        yield f"search_standard_overlapping_{cid}", "standard", "overlapping", kw, collection("AC_STANDARD_OVERLAPPING")
    for mk, coll in (("standard", "AC_STANDARD_ANCHORED_NON_OVERLAPPING"),
                     ("leftmost_first", "AC_LEFTMOST_FIRST_ANCHORED"),
                     ("leftmost_longest", "AC_LEFTMOST_LONGEST_ANCHORED")):
        for cid, kw in ANCHORED_CONFIGS:
            yield f"search_{mk}_anchored_{cid}", mk, "anchored", kw, collection(coll)
    # ASCII case insensitivity src/tests.rs:1182-1323
    ci, ci_no, ci_ov = ["ASCII_CASE_INSENSITIVE"], ["ASCII_CASE_INSENSITIVE", "ASCII_CASE_INSENSITIVE_NON_OVERLAPPING"], \
        ["ASCII_CASE_INSENSITIVE", "ASCII_CASE_INSENSITIVE_OVERLAPPING"]
    c = {"ascii_case_insensitive": True}
    yield "acasei_standard_default", "standard", "find_iter", dict(c, prefilter=False), _groups(ci)
    yield "acasei_standard_nfa_noncontig_default", "standard", "find_iter", dict(c, kind="nnfa", prefilter=False), _groups(ci)
    yield "acasei_standard_nfa_contig_default", "standard", "find_iter", dict(c, kind="cnfa", prefilter=False), _groups(ci)
    yield "acasei_standard_dfa_default", "standard", "find_iter", dict(c, kind="dfa"), _groups(ci_no)
    for cid, k in (("default", None), ("nfa_noncontig_default", "nnfa"), ("nfa_contig_default", "cnfa"), ("dfa_default", "dfa")):
        kw = dict(c) if k is None else dict(c, kind=k)
        yield f"acasei_standard_overlapping_{cid}", "standard", "overlapping", kw, _groups(ci_ov)
        yield f"acasei_leftmost_first_{cid}", "leftmost_first", "find_iter", kw, _groups(ci_no)
        yield f"acasei_leftmost_longest_{cid}", "leftmost_longest", "find_iter", kw, _groups(ci_no)


def doctests():
    return VECTORS["doctests"]


def unhex(v):
    return [bytes.fromhex(p) for p in v["patterns"]], bytes.fromhex(v["haystack"]), [tuple(m) for m in v["matches"]]
