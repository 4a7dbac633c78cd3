#!/usr/bin/env python3
"""Collapse rocprofv3 --pmc counter_collection CSVs into per-kernel averages (one line per kernel/counter)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if "pygim" not in k:
                continue
            name = k.split("(")[0].replace("void pygim::", "")
            key = (name, row["Counter_Name"])
            agg[key][0] += float(row["Counter_Value"])
            agg[key][1] += 1
lines = []
for (k, c), (tot, n) in sorted(agg.items()):
    lines.append(f"{k:40s} {c:34s} avg/dispatch {tot / n:18.1f}  dispatches {n}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(root, "summary.txt"), "w").write(txt + "\n")
