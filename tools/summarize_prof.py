#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + per-dispatch PMC rows) into a small markdown summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print(f"# rocprofv3 summary ({out})\n")
for f in find("stats/**/*kernel_stats.csv"):
    print(f"## kernel stats ({os.path.relpath(f, out)})\n")
    rows = list(csv.DictReader(open(f)))
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        name = r["Name"].split("(")[0][:70]
        print(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | "
              f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    print()
for d in find("pmc_*"):
    cname = os.path.basename(d)[4:]
    files = [f for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)]
    if not files:
        print(f"## {cname}: no counter_collection.csv\n")
        continue
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != cname:
                continue
            k = r["Kernel_Name"].split("(")[0][:70]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    print(f"## {cname} per dispatch (mean over dispatches)\n")
    print("| kernel | dispatches | mean value |")
    print("|---|---|---|")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {n} | {v / n:.1f} |")
    print()
