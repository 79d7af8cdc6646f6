#!/usr/bin/env python3
"""Tuning aid: the last N dispatches of a rocprofv3 --kernel-trace csv -- name, duration, gap to the previous dispatch.
usage: trace_tail.py <dir with *kernel_trace.csv> [N=200] > out"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%-44s dur %8.2f us  gap %7.2f us  grid %s wg %s lds %s vgpr %s" % (r["Kernel_Name"][:44], (e - s) / 1e3, gap, r.get("Grid_Size"), r.get("Workgroup_Size"),
                                                                           r.get("LDS_Block_Size"), r.get("VGPR_Count")))
    prev = e
