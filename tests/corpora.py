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
