#!/usr/bin/env python3
"""Call latency against haystack size -- the reference benchmarks haystacks of 10 KB - 1.5 MB (benchmarks/haystacks/), the
numbers of bench.py are at 256 MiB - 8 GiB.  For two pattern sets (the headline's 1 000 random patterns over random ASCII;
five names over sherlock.txt) and sizes 4 KiB ... 1 GiB: the time of one completed `find_overlapping_iter` and one
`find_iter` (LeftmostFirst) call, haystack and records resident on the device, and the same calls with haystack and records
on the host (PCIe inside the call) -- beside the oracle's restatement of the reference loops on ONE host core (the checker,
timed as the CPU baseline; it takes no part in the GPU numbers).  The last line of each set names the crossover: the smallest
size from which the GPU call is faster than the CPU loop.
usage: bench_latency.py [max_mib]      -> one JSON line per (set, size)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aho_corasick_amd as ac
import corpora
from oracle import orc   # (CPU baseline only)

max_bytes = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
sizes = [s for s in (4 << 10, 16 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20, 16 << 20, 64 << 20, 256 << 20, 1 << 30) if s <= max_bytes]


def med(f, reps):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2]


def sets():
    pats = ac.gen_patterns(1000, seed=0xAC01)
    yield "1000 random 4-16 B patterns / random ASCII (bench.py's headline set)", pats, lambda n: orc.gen_haystack(0, n, seed=0xAC02)
    names = [b"Sherlock Holmes", b"John Watson", b"Irene Adler", b"Inspector Lestrade", b"Professor Moriarty"]
    yield "five names / sherlock.txt tiled (the reference's curated/sherlock)", names, lambda n: corpora.haystack("sherlock.txt", n)


for label, pats, make in sets():
    a_ov = ac.AhoCorasick.builder().build(pats)
    a_lf = ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).build(pats)
    o_ov, o_lf = orc.Oracle(pats), orc.Oracle(pats, match_kind=1)
    out = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    cross = {"ov_device": None, "ov_host": None, "lf_device": None, "lf_host": None}
    for n in sizes:
        hay = make(n)
        dev = torch.from_numpy(hay).cuda()
        reps = 30 if n <= (16 << 20) else 7
        for _ in range(3):   # (upload, scratch, clocks)
            a_ov.overlapping_device(dev, out=out); a_lf.find_iter_device(dev, out)
            if n <= (256 << 20):
                a_ov.find_overlapping_iter(hay, as_numpy=True); a_lf.find_iter(hay, as_numpy=True)
        torch.cuda.synchronize()
        row = {"set": label, "bytes": n,
               "ov_device_us": med(lambda: a_ov.overlapping_device(dev, out=out), reps) * 1e6,
               "lf_device_us": med(lambda: a_lf.find_iter_device(dev, out), reps) * 1e6}
        if n <= (256 << 20):
            row["ov_host_us"] = med(lambda: a_ov.find_overlapping_iter(hay, as_numpy=True), reps) * 1e6
            row["lf_host_us"] = med(lambda: a_lf.find_iter(hay, as_numpy=True), reps) * 1e6
        cn = min(n, 64 << 20)   # (the CPU loop is timed on at most 64 MiB and scaled: it is linear in the haystack)
        creps = 5 if cn > (1 << 20) else 30
        row["cpu_ov_us"] = med(lambda: o_ov.find_overlapping_iter(hay[:cn], as_numpy=True), creps) * 1e6 * (n / cn)
        row["cpu_lf_us"] = med(lambda: o_lf.find_iter(hay[:cn], as_numpy=True), creps) * 1e6 * (n / cn)
        row["matches"] = int(a_ov.overlapping_device(dev, out=out)[0])
        for k in ("ov_device", "ov_host", "lf_device", "lf_host"):
            if k + "_us" in row:
                faster = row[k + "_us"] < row["cpu_" + k[:2] + "_us"]
                if faster and cross[k] is None:
                    cross[k] = n
                if not faster:
                    cross[k] = None
        print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
        del dev
    print(json.dumps({"set": label, "gpu_call_faster_than_one_cpu_core_from_bytes": cross,
                      "cpu": "oracle (C restatement of the reference loops, -O3), one core"}), flush=True)
