#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSV output per kernel name (tuning aid)."""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "mc_gemv"
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(rows.items()):
    if pat not in k:
        continue
    n = max(len(v) for v in cs.values())
    print(k, "dispatches", n)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):14.1f}")
