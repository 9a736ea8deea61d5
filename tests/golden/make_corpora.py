#!/usr/bin/env python3
"""Copies the reference's own benchmark inputs (reusable test DATA, not source code) into
tests/golden/corpora/ as gzip files, with a manifest of their SHA-256 sums.  Run in the build container, where
/root/reference exists; the GPU box only sees the committed fixtures.

    sherlock.txt, opensubtitles/en-huge.txt   haystacks of the reference's `curated` / `opensubtitles` benchmark groups
                                              (benchmarks/definitions/*.toml)
    words-100 / words-5000 / words-15000      its dictionary pattern sets (benchmarks/regexes/)
    dictionary-15                             dictionary/english/length-15.txt, the `dictionary-15` curated benchmark

They are the inputs on which a 4-byte prefix filter behaves differently from random ASCII: natural text searched for
dictionary words."""
import gzip
import hashlib
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "corpora")
FILES = {
    "sherlock.txt": "benchmarks/haystacks/sherlock.txt",
    "en-huge.txt": "benchmarks/haystacks/opensubtitles/en-huge.txt",
    "words-100": "benchmarks/regexes/words-100",
    "words-5000": "benchmarks/regexes/words-5000",
    "words-15000": "benchmarks/regexes/words-15000",
    "dictionary-15": "benchmarks/regexes/dictionary/english/length-15.txt",
}

os.makedirs(OUT, exist_ok=True)
manifest = {}
for name, rel in FILES.items():
    data = open(os.path.join(REF, rel), "rb").read()
    with open(os.path.join(OUT, name + ".gz"), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0, compresslevel=9) as g:   # mtime=0: reproducible bytes
            g.write(data)
    manifest[name] = {"source": rel, "bytes": len(data), "sha256": hashlib.sha256(data).hexdigest()}
json.dump(manifest, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(manifest, indent=1))
