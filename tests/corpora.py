"""The reference's own benchmark inputs as committed fixtures (tests/golden/corpora/, made by tests/golden/make_corpora.py)."""
import gzip
import hashlib
import json
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corpora")
MANIFEST = json.load(open(os.path.join(_DIR, "MANIFEST.json")))


def raw(name):
    data = gzip.open(os.path.join(_DIR, name + ".gz"), "rb").read()
    m = MANIFEST[name]
    assert len(data) == m["bytes"] and hashlib.sha256(data).hexdigest() == m["sha256"], name
    return data


def words(name):
    """Pattern list of a words-N file: one pattern per line (benchmarks/regexes/words-N)."""
    return [w for w in raw(name).split(b"\n") if w]


def haystack(name, size=None):
    """numpy uint8 copy of a haystack file, tiled up to `size` bytes when given."""
    a = np.frombuffer(raw(name), dtype=np.uint8)
    if size is None:
        return a.copy()
    reps = -(-size // len(a))
    return np.tile(a, reps)[:size].copy()


def bench_defs():
    """The reference's benchmark definitions (benchmarks/definitions/*.toml) as extracted by make_corpora.py:
    {definition file: [bench, ...]}."""
    return json.load(open(os.path.join(_DIR, "bench_defs.json")))


def bench_patterns(b):
    if "patterns_file" in b:
        return words(b["patterns_file"])
    return [bytes.fromhex(p) for p in b["patterns_hex"]]


def bench_haystack(b):
    if "haystack_file" in b:
        return haystack(b["haystack_file"])
    h = b["haystack"]
    return np.frombuffer((h["contents"] * h["repeat"] + h["append"]).encode("utf-8"), dtype=np.uint8).copy()


def bench_expected(b, engine):
    """The reference's expected match count of bench b for one of its engine names (count may be a list of
    {engine regex, count}: the first regex that matches wins), or None if the bench does not run that engine."""
    import re
    if engine not in b["engines"]:
        return None
    c = b["count"]
    if isinstance(c, int):
        return c
    for e in c:
        if re.fullmatch(e["engine"], engine):
            return e["count"]
    return None
