#!/usr/bin/env python3
"""Kernels + memory copies of a rocprofv3 trace on one time line: prints the last N events with gaps."""
import csv, glob, sys, re
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 80
ev = []
for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + re.sub(r"\(.*", "", r["Kernel_Name"])[:50]))
for p in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))[:40]))
ev.sort()
ev = ev[-n:]
t0, prev = ev[0][0], ev[0][0]
for s, e, name in ev:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} gap {(s - prev) / 1e3:7.1f}  {name}")
    prev = max(prev, e)
