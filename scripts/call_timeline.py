#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of scripts/trace_defs.py + its stdout -> one timeline per measured call: every launch with
its offset from the call's first launch, duration and the idle gap in front of it (host round trips show as gaps).
usage: call_timeline.py <dir with *kernel_trace.csv> <trace_defs stdout> [rep to print, default 2]"""
import csv, glob, json, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
calls = [json.loads(l) for l in open(sys.argv[2]) if l.startswith("{")]
want_rep = int(sys.argv[3]) if len(sys.argv) > 3 else 2


def is_marker(r):
    n = r["Kernel_Name"]
    return "at::native" in n or "elementwise_kernel" in n


# segments: the launches between one marker and the next marker (or the end); warm-up calls sit in front of the first
# marker of a bench and are cut off because a segment ends at the first gap > 700 us (the sleep between calls)
segs = []
i = 0
while i < len(rows):
    if is_marker(rows[i]):
        j = i + 1
        seg = []
        while j < len(rows) and not is_marker(rows[j]):
            if seg and int(rows[j]["Start_Timestamp"]) - int(seg[-1]["End_Timestamp"]) > 700_000:
                break
            seg.append(rows[j]); j += 1
        segs.append(seg)
        i = j
    else:
        i += 1
if len(segs) != len(calls):
    print(f"warning: {len(segs)} trace segments, {len(calls)} calls on stdout")
for c, seg in zip(calls, segs):
    if c["rep"] != want_rep or not seg:
        continue
    t0 = int(seg[0]["Start_Timestamp"])
    span = (int(seg[-1]["End_Timestamp"]) - t0) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
    print(f"=== {c['bench']} [{c['kind']}] records {c['records']} wall {c['wall_us']} us ({c['GBps']} GB/s); "
          f"{len(seg)} launches, first start to last end {span:.1f} us, busy {busy:.1f} us")
    prev = None
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        name = r["Kernel_Name"].replace("acgpu::", "").replace("(anonymous namespace)::", "")[:70]
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:6.1f}  {name}")
        prev = e
