#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch.
usage: pmc_summary.py <dir-with-pass-subdirs> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        k = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[k] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (d, c), v in per_dispatch.items():
        agg[names[d].split("(")[0][-60:]][c].append(v)
for k in sorted(agg):
    if filt in k:
        print(k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print(f"   {c:28s} mean {sum(v) / len(v):18.1f}  n={len(v)}")
