#!/usr/bin/env python3
"""Throughput over the reference's own benchmark definitions (tests/golden/corpora/bench_defs.json): every bench's
pattern set over its haystack tiled to 256 MiB, default engine: overlapping search (whole call, count kernel) and
find_iter under LeftmostFirst -- every call COMPLETED (the record buffer is sized from the call's own count) -- and, beside
them, the CPU baseline of the same search: the oracle's restatement of the reference loops on ONE host core over a
bounded sample of the same haystack (the checker timed as a baseline, like bench.py's cpu_baseline leg; it takes no part
in the GPU numbers).  One JSON line per bench + one summary line per definition file.
usage: bench_defs.py [mib] [engine] [name filter,...] [variant=value,...]   (BENCH_DEFS_NO_CPU=1 skips the CPU side)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
import corpora
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
engine = sys.argv[2] if len(sys.argv) > 2 else "auto"
only = [x for x in sys.argv[3].split(",") if x] if len(sys.argv) > 3 and sys.argv[3] else None
variants = dict((v.split("=")[0], int(v.split("=")[1])) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else {}


def with_variants(b):
    for k_, v_ in variants.items():
        b.gpu_variant(k_, v_)
    return b
out = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
NO_CPU = os.environ.get("BENCH_DEFS_NO_CPU") == "1"
CPU_SAMPLE = 8 << 20


def fit(records):
    """the shared record buffer, grown to hold `records` (24 bytes each)"""
    global out
    if out.numel() < records * 24:
        out = None
        torch.cuda.empty_cache()
        out = torch.empty(int(records * 24 * 1.02) + 4096, dtype=torch.uint8, device="cuda")


def cpu_side(pats, hay_np):
    """one host core, the oracle: overlapping (Standard) and find_iter (LeftmostFirst) over the first CPU_SAMPLE bytes"""
    from oracle import orc   # (the checker, timed as the CPU baseline)
    h = np.ascontiguousarray(hay_np[:CPU_SAMPLE])
    res = {}
    for name, mk, run in (("cpu_ov_GBps", 0, lambda o: o.find_overlapping_iter(h, as_numpy=True)),
                          ("cpu_lf_GBps", 1, lambda o: o.find_iter(h, as_numpy=True))):
        o = orc.Oracle(pats, match_kind=mk)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); r = run(o); ts.append(time.perf_counter() - t)
        res[name] = round(len(h) / sorted(ts)[1] / 1e9, 4)
    res["cpu_sample_mib"] = len(h) >> 20
    return res
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
        a = with_variants(ac.AhoCorasick.builder().gpu_engine(engine)).build(pats)
        p = _lib.CProfile()
        tiled = np.tile(hay, -(-n // len(hay)))[:n]
        try:
            m, ok = a.overlapping_device(d, out=out, profile=p)
            if not ok:
                fit(m)
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
                   "ov_kernel_GBps": round(n / (float(np.mean(ks)) or 1e9) / 1e6, 1),
                   "ov_records_per_s_G": round(int(m) / dt / 1e9, 3), "ov_write_GBps": round(int(m) * 24 / dt / 1e9, 1)}
        except Exception as e:
            row = {"family": family, "bench": b["name"], "error": str(e)[:80]}
        try:
            lf = with_variants(ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst)).build(pats)
            k, okl = lf.find_iter_device(d, out)
            if not okl:
                fit(k)
            k, okl = lf.find_iter_device(d, out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                k, okl = lf.find_iter_device(d, out)
            torch.cuda.synchronize()
            row["lf_find_iter_GBps"] = round(n / ((time.perf_counter() - t0) / 3) / 1e9, 1)
            row["lf_matches"] = int(k)
            row["lf_ok"] = bool(okl)
        except Exception as e:
            row["lf_error"] = str(e)[:80]
        if not NO_CPU:
            try:
                row.update(cpu_side(pats, tiled))
            except Exception as e:
                row["cpu_error"] = str(e)[:80]
        rows.append(row)
        print(json.dumps(row), flush=True)
        del d
    if not rows:
        continue
    ov = [r["ov_call_GBps"] for r in rows if "ov_call_GBps" in r]
    lf = [r["lf_find_iter_GBps"] for r in rows if "lf_find_iter_GBps" in r]
    print(json.dumps({"family": family, "benches": len(rows), "ov_call_GBps_min_med_max": [min(ov), float(np.median(ov)), max(ov)] if ov else None,
                      "lf_find_iter_GBps_min_med_max": [min(lf), float(np.median(lf)), max(lf)] if lf else None}), flush=True)
