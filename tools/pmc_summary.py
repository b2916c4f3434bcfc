#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 counter CSVs under a directory (one sub-directory per pass)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    n = max(len(x) for x in v.values())
    tot = sum(sum(x) for x in v.values())
    if "k_mega" in k or "k_wf" in k:
        print(k, "dispatches", n)
        for c in sorted(v):
            print(f"  {c:32s} {sum(v[c]) / len(v[c]):16.0f}")
