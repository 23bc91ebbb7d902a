#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: launches, summed and mean counter value.
usage: pmc_summary.py <dir with *counter_collection.csv> <out.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
agg = defaultdict(lambda: [0, 0.0])
dispatches = defaultdict(set)
for f in files:
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            key = (row["Kernel_Name"].split("(")[0][:80], row["Counter_Name"])
            agg[key][1] += float(row["Counter_Value"])
            dispatches[key].add(row["Dispatch_Id"])
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum", "Mean_per_dispatch"])
    for (k, c), (_, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        n = len(dispatches[(k, c)])
        w.writerow([k, c, n, f"{s:.1f}", f"{s / max(n, 1):.1f}"])
print("wrote", out, "from", files)
