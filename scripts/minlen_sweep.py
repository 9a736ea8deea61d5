#!/usr/bin/env python3
"""The shortest-pattern cliff of the natural-text path: the reference's words-5000 list (shortest word: 9 bytes) plus ONE
extra word of k = 8 .. 3 bytes over sherlock.txt tiled to 1 GiB, default engine: which kernel serves the search and what a
step costs.  One JSON line per row; `vs_all_long` = step time relative to the unmodified list.
usage: minlen_sweep.py [gib]"""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
import corpora

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = int(gib * (1 << 30))
text = corpora.haystack("sherlock.txt")
nat = torch.from_numpy(np.tile(text, -(-n // len(text)))[:n].copy()).cuda()
words = corpora.words("words-5000")
out = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
extra = {None: None, 8: b"daughter", 7: b"morning", 6: b"window", 5: b"chair", 4: b"lamp", 3: b"cab"}
base_ms = None
for k, w in extra.items():
    pats = list(words) + ([w] if w else [])
    a = ac.AhoCorasick.builder().build(pats)
    p = _lib.CProfile()
    for _ in range(4):
        m, ok = a.overlapping_device(nat, out=out, profile=p)
    torch.cuda.synchronize()
    ks, t0 = [], time.perf_counter()
    for _ in range(8):
        m, ok = a.overlapping_device(nat, out=out, profile=p)
        ks.append(p.ms_scan)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 8 * 1e3
    crc = zlib.crc32(out[: int(m) * 24].cpu().numpy().tobytes()) if ok else None
    # the same records from the transition walk (engine "walk"), once
    b = ac.AhoCorasick.builder().gpu_engine("walk").build(pats)
    m2, ok2 = b.overlapping_device(nat, out=out)
    crc2 = zlib.crc32(out[: int(m2) * 24].cpu().numpy().tobytes()) if ok2 else None
    if base_ms is None:
        base_ms = step
    print(json.dumps({"extra_word": w.decode() if w else None, "min_len": min(map(len, pats)), "patterns": len(pats), "gib": gib,
                      "engine": int(p.engine_used), "routed": int(p.routed), "matches": int(m), "kernel_ms": round(float(np.mean(ks)), 4),
                      "step_ms": round(step, 4), "step_GBps": round(n / step / 1e6, 1), "vs_all_long": round(step / base_ms, 3),
                      "same_records_as_walk": bool(ok and ok2 and m == m2 and crc == crc2)}), flush=True)
    del a, b
