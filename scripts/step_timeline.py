#!/usr/bin/env python3
"""Timeline of the launches between two scans: rocprofv3 --kernel-trace CSV -> for the LAST interval between two launches of
the kernel named by argv[2], every launch with its start offset and duration (us), plus the idle time between launches.
usage: step_timeline.py <dir with *kernel_trace.csv> <scan kernel substring>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
scans = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
if len(scans) < 3:
    sys.exit("not enough scans")
for which in (-3, -2):
    a, b = scans[which], scans[which + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = None
    busy = 0
    print(f"--- interval {which}: {b - a} launches, scan to scan {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        name = r["Kernel_Name"].replace("acgpu::", "").replace("(anonymous namespace)::", "")[:60]
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:6.1f}  {name}")
        busy += e - s
        prev_end = e
    print(f"    idle before the next scan: {(int(rows[b]['Start_Timestamp']) - prev_end) / 1e3:.1f} us")
