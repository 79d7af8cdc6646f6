#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel.

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly HALF of the bytes of a
wide coalesced streaming read (16 B per lane), so it is doubled; WRITE_SIZE reads true."""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch):
    if not k.startswith("mc_"):
        continue
    f = sum(fetch[k]) / len(fetch[k]) * 1024.0 * 2.0
    w = (sum(write[k]) / len(write[k]) * 1024.0) if k in write else 0.0
    out[k] = dict(launches=len(fetch[k]), fetch_bytes_per_launch=round(f), write_bytes_per_launch=round(w),
                  hbm_bytes_per_launch=round(f + w))
import os
head = None
try:
    head = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".build_head")).read().strip()
except OSError:
    pass
print(json.dumps(dict(git_head=head, note="FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count correction) + WRITE_SIZE KiB x 1024",
                      kernels=out), indent=1))
