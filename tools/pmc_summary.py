#!/usr/bin/env python3
"""Per-kernel mean of each counter from a rocprofv3 counter_collection CSV."""
import collections
import csv
import sys

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(path) as fh:
    for row in csv.DictReader(fh):
        k = row.get("Kernel_Name", "")
        if pat and pat not in k:
            continue
        acc[k.split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s mean %.6g  (n=%d, min %.6g, max %.6g)" % (c, sum(v) / len(v), len(v), min(v), max(v)))
