#!/usr/bin/env python3
"""In-flow timeline of one forward out of a rocprofv3 kernel trace of bench.py: per-kernel duration
and the idle gap before it, averaged over the forwards found (conv head -> ... -> mse_kernel).
    python tools/flow_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, re, sys
from collections import defaultdict

rows = []
for p in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1:]:
    with open(p) as f:
        for r in csv.DictReader(f):
            if "aft::" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])))
rows.sort()
ends = [i for i, r in enumerate(rows) if "mse_kernel" in r[2]]
flows = []
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1: b + 1]
    if sum("attn_kernel" in r[2] for r in seg) == 6 and len(seg) <= 24:
        flows.append(seg)
print(f"{len(flows)} complete forwards")
agg = defaultdict(lambda: [0, 0.0, 0.0])
spans = []
for seg in flows:
    spans.append((seg[-1][1] - seg[0][0]) / 1e3)
    prev = None
    for i, (s, e, n) in enumerate(seg):
        key = f"{i:02d} {n[:60]}"
        agg[key][0] += 1
        agg[key][1] += (e - s) / 1e3
        agg[key][2] += 0.0 if prev is None else max(0, s - prev) / 1e3
        prev = e
print(f"span first-start..last-end: mean {sum(spans) / len(spans):.1f} us  min {min(spans):.1f} us")
tk = tg = 0
for k in sorted(agg):
    c, t, g = agg[k]
    tk += t / c; tg += g / c
    print(f"  {k:<64} dur={t / c:8.2f} us  gap_before={g / c:6.2f} us")
print(f"  sum of kernel durations {tk:.1f} us, sum of gaps {tg:.1f} us")
