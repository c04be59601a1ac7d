#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: mean of every counter over the dispatches.
Usage: python scripts/pmc_kernel.py <dir> [substring]"""
import collections
import csv
import glob
import os
import sys

root, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub in k:
            acc[k.split("(")[0][-70:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, "dispatches", len(next(iter(cs.values()))))
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v) / len(v):16.1f}")
