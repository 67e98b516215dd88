#!/usr/bin/env python3
"""Print a compact timeline of the last frame in a rocprofv3 kernel trace CSV."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0][:28], r.get("Queue_Id", "?")))
rows.sort()
# last N kernels
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
last = rows[-n:]
t0 = last[0][0]
for s, e, name, q in last:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{q}  {name}")
print("span us", (max(e for _, e, _, _ in last) - t0) / 1e3)
