#!/usr/bin/env python3
"""Throughput over the reference's own benchmark definitions (tests/golden/corpora/bench_defs.json): every bench's
pattern set over its haystack tiled to 256 MiB, default engine: overlapping search (whole call, count kernel) and
find_iter under LeftmostFirst.  One JSON line per bench + one summary line per definition file.  usage: bench_defs.py [mib]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
import corpora
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
engine = sys.argv[2] if len(sys.argv) > 2 else "auto"
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
out = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
defs = corpora.bench_defs()
extra = {"dictionary": [{"name": "sorted.txt (123 115 words) / sherlock", "patterns_file": "dictionary-sorted", "haystack_file": "sherlock.txt"},
                        {"name": "length-10 / en-sampled", "patterns_file": "dictionary-10", "haystack_file": "en-sampled.txt"}]}
for family, benches in list(defs.items()) + list(extra.items()):
    rows = []
    for b in benches:
        if only and not any(o in b["name"] for o in only):
            continue
        pats, hay = corpora.bench_patterns(b), corpora.bench_haystack(b)
        d = torch.from_numpy(np.tile(hay, -(-n // len(hay)))[:n].copy()).cuda()
        a = ac.AhoCorasick.builder().gpu_engine(engine).build(pats)
        p = _lib.CProfile()
        try:
            for _ in range(2):
                m, ok = a.overlapping_device(d, out=out, profile=p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ks = []
            for _ in range(3):
                m, ok = a.overlapping_device(d, out=out, profile=p); ks.append(p.ms_scan)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            row = {"family": family, "bench": b["name"], "patterns": len(pats), "min_len": min(map(len, pats)), "matches": int(m), "ok": bool(ok),
                   "engine": int(p.engine_used), "routed": int(p.routed), "ov_call_GBps": round(n / dt / 1e9, 1),
                   "ov_kernel_GBps": round(n / (float(np.mean(ks)) or 1e9) / 1e6, 1)}
        except Exception as e:   # (a saturated result that does not fit the output buffer)
            row = {"family": family, "bench": b["name"], "error": str(e)[:80]}
        try:
            lf = ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).build(pats)
            lf.find_iter_device(d, out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                k = lf.find_iter_device(d, out)[0]
            torch.cuda.synchronize()
            row["lf_find_iter_GBps"] = round(n / ((time.perf_counter() - t0) / 3) / 1e9, 1)
            row["lf_matches"] = int(k)
        except Exception as e:
            row["lf_error"] = str(e)[:80]
        rows.append(row)
        print(json.dumps(row), flush=True)
        del d
    if not rows:
        continue
    ov = [r["ov_call_GBps"] for r in rows if "ov_call_GBps" in r]
    lf = [r["lf_find_iter_GBps"] for r in rows if "lf_find_iter_GBps" in r]
    print(json.dumps({"family": family, "benches": len(rows), "ov_call_GBps_min_med_max": [min(ov), float(np.median(ov)), max(ov)] if ov else None,
                      "lf_find_iter_GBps_min_med_max": [min(lf), float(np.median(lf)), max(lf)] if lf else None}), flush=True)
