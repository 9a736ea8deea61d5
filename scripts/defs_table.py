#!/usr/bin/env python3
"""One line per row of a bench_defs.py output file."""
import json, sys
for l in open(sys.argv[1]):
    if not l.startswith("{"):
        continue
    r = json.loads(l)
    if "bench" not in r:
        continue
    if "error" in r:
        print(f"{r['family']:12s} {r['bench']:36s} ERROR {r['error']}")
        continue
    print(f"{r['family']:12s} {r['bench']:36s} P={r['patterns']:6d} M={r['matches']:10d} eng={r['engine']} routed={r['routed']} "
          f"kern={r['ov_kernel_GBps']:7.1f} call={r['ov_call_GBps']:7.1f} rec/s={r.get('ov_records_per_s_G', 0):6.2f}G lf={r.get('lf_find_iter_GBps', 0):7.1f} ok={r['ok']}/{r.get('lf_ok')}")
