#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 counter_collection.csv (one row per dispatch and counter)."""
import csv
import sys
from collections import defaultdict

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(path)):
    k = row.get("Kernel_Name", "")
    if pat and pat not in k:
        continue
    key = (k.split("(")[0][:90], row["Counter_Name"])
    acc[key][0] += float(row["Counter_Value"])
    acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:90s} {c:28s} mean/dispatch {v / n:18.1f}  dispatches {n}")
