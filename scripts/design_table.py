#!/usr/bin/env python3
"""DESIGN.md section 4's table rows from a bench.py line (profiles/rNN_bench.json): usage design_table.py <bench.json>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def row(name, kernel, ms, value, r, cpu=""):
    tr = f"{r['traffic'] / 1e9:.2f} GB ({r['traffic'] / r['algorithmic_bytes_per_launch']:.2f} x)" if r.get("traffic") else "-"
    print(f"| {name} | `{kernel}` | {ms:.3f} | {value:,.0f} | {r['kernel_ms']:.3f} | {r['achieved']:,.0f} = {100 * r['frac']:.1f} % | {tr} | {cpu} |".replace(",", " "))


r = d["roofline"]
c = d.get("cpu_baseline") or {}
row("**headline**: 1 000 patterns, 8 GiB, overlapping, default engine", "k_pf_count<false,false>", d["ms_per_step"], d["value"], r,
    f"{c.get('value')} GB/s ({c.get('all_cores', {}).get('threads')} threads: {c.get('all_cores', {}).get('value')})")
print(f"  (empirical peak {r.get('empirical_peak')}, frac of it {r.get('frac_of_empirical')})")
for k, e in d.get("engines", {}).items():
    row(f"`engines.{k}`", e.get("kernel", "?"), e["ms_per_step"], e["value"], e)
for a in d.get("also", []):
    if "error" in a:
        print("| ERROR", a)
        continue
    cpu = a.get("cpu_baseline") or {}
    extra = ""
    if "enqueue_form" in a:
        extra = f" (enqueue-only {a['enqueue_form']['ms_per_step']:.3f} ms, {a['enqueue_form']['value']:.0f} GB/s)"
    if "find_iter" in a:
        extra = f" (find_iter {a['find_iter']['ms_per_step']:.3f} ms, {a['find_iter']['value']:.0f} GB/s; {a['matches']} records)"
    row(a["workload"][:70] + extra, a["roofline"].get("kernel", "?"), a["ms_per_step"], a["value"], a["roofline"], str(cpu.get("value", "")))
for cst in d.get("costs", []):
    print("cost:", cst)
