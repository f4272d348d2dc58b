#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv:  tools/pmc_summary.py <dir> [substr ...]"""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
want = sys.argv[2:]
f = max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if want and not any(w in k for w in want):
        continue
    agg[k.split("(")[0][-70:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in agg.items()}
print(json.dumps(out, indent=1))
