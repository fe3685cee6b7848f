#!/usr/bin/env python3
"""rocprofv3 --pmc CSV output directory -> markdown table: per kernel, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "kdb::" not in r["Kernel_Name"]:
            continue
        if ("scatter_bases_kernel" in r["Kernel_Name"] or "count_smallk_kernel" in r["Kernel_Name"]) and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 8000:
            continue          # (the variant compiled for the other batch shape: it returns at once)
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kdb::", "")[:60]
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
ctrs = sorted({c for v in agg.values() for c in v})
print(f"### {os.path.basename(d)}: mean per dispatch\n")
print("| kernel | dispatches | " + " | ".join(ctrs) + " |")
print("|---|---|" + "---|" * len(ctrs))
for k, v in sorted(agg.items()):
    n = max(x[0] for x in v.values())
    print(f"| {k} | {n} | " + " | ".join(f"{v[c][1] / max(v[c][0], 1):.4g}" if c in v else "-" for c in ctrs) + " |")
