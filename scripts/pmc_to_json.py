#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (scripts/gpu_pmc.sh output dir) into one JSON of per-dispatch averages
for one kernel.  usage: pmc_to_json.py <gpurun_out/pmc_dir> <kernel substring> <out.json> [note] [algorithmic GiB per launch]"""
import csv, glob, json, sys, collections

d, kern, out = sys.argv[1:4]
note = sys.argv[4] if len(sys.argv) > 4 else ""
gib = float(sys.argv[5]) if len(sys.argv) > 5 else None
res = {}
names, ndisp = set(), 0
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if kern in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Counter_Name"]].add(r.get("Dispatch_Id"))
            kn = r["Kernel_Name"]
            at = kn.find(kern)
            end = kn.find("(", at)
            names.add(kn[at:end if end > 0 else len(kn)].strip())   # e.g. k_pf_count<false, false>
    for k, v in tot.items():
        res[k] = v / max(1, len(disp[k]))
        ndisp = max(ndisp, len(disp[k]))
rd = {k: res.get(f"TCC_EA0_RDREQ_{k}_sum") for k in ("32B", "64B", "128B")}
if all(v is not None for v in rd.values()):
    res["hbm_read_bytes"] = rd["32B"] * 32 + rd["64B"] * 64 + rd["128B"] * 128
res["_kernel"] = sorted(names)[0] if names else None   # the kernel (template arguments included) the counters belong to
res["_kernels_matched"] = sorted(names)
res["_dispatches"] = ndisp
if gib is not None:
    res["_gib"] = gib
res["_note"] = note or f"per-dispatch averages of {kern}; separate rocprofv3 --pmc passes (scripts/gpu_pmc.sh)"
json.dump(dict(sorted(res.items())), open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
