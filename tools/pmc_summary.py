#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name over its dispatches."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    if not any(k in name for k in ("render", "spheres", "traverse", "shade", "path")):
        continue
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]
        print(f"   {c:32s} mean {sum(v) / len(v):16.1f}  n={len(v)}")
