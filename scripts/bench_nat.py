#!/usr/bin/env python3
"""Natural text against the reference's dictionaries (its own benchmark corpora): whole call and kernel times.
usage: bench_nat.py [steps] [haystack name filter]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
import corpora
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = 1 << 30
out = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
only = sys.argv[2] if len(sys.argv) > 2 else None
for hay_name, words_name in (("sherlock.txt", "words-5000"), ("en-huge.txt", "words-15000"), ("sherlock.txt", "words-100")):
    if only and (only not in hay_name or words_name == "words-100"):
        continue
    text = corpora.haystack(hay_name)
    nat = torch.from_numpy(np.tile(text, -(-n // len(text)))[:n].copy()).cuda()
    b = ac.AhoCorasick.builder().match_kind(ac.MatchKind.Standard)
    for kv in os.environ.get("NAT_VARIANTS", "").split(","):   # e.g. NAT_VARIANTS=pfx_tails=1,eo_fused=0
        if kv:
            b.gpu_variant(kv.split("=")[0], int(kv.split("=")[1]))
    a = b.build(corpora.words(words_name))
    p = _lib.CProfile()
    for _ in range(3):
        m, ok = a.overlapping_device(nat, out=out, profile=p)
    torch.cuda.synchronize()
    ks, cs, fs, t0 = [], [], [], time.perf_counter()
    for _ in range(steps):
        m, ok = a.overlapping_device(nat, out=out, profile=p)
        ks.append(p.ms_scan); cs.append(p.ms_compact); fs.append(p.ms_fill)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"haystack": hay_name, "words": words_name, "matches": int(m), "engine": int(p.engine_used), "routed": int(p.routed),
                      "call_ms": round(dt * 1e3, 3), "call_GBps": round(n / dt / 1e9, 1), "kernel_ms": round(float(np.mean(ks)), 3),
                      "kernel_GBps": round(n / float(np.mean(ks)) / 1e6, 1), "rank_ms": round(float(np.mean(cs)), 3),
                      "emit_ms": round(float(np.mean(fs)), 3)}), flush=True)
    del nat, a
