#!/usr/bin/env python3
"""Per-kernel breakdown of ONE steady-state training step out of a rocprofv3 kernel trace of
tools/train_bench.py: the dispatches between the last two adam_kernel launches (MIOpen's first-call
algorithm search and the warm-up are excluded that way).
    python tools/train_step_breakdown.py <dir containing *_kernel_trace.csv>"""
import csv, glob, os, re, sys
from collections import defaultdict

paths = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
if len(adam) < 2:
    sys.exit("trace holds fewer than two optimizer steps")
win = rows[adam[-2] + 1: adam[-1] + 1]
agg = defaultdict(lambda: [0, 0.0])
for s, e, n in win:
    n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))
    n = re.sub(r"^void ", "", n)[:72]
    agg[n][0] += 1
    agg[n][1] += (e - s) / 1e3
span = (win[-1][1] - win[0][0]) / 1e3
busy = sum(v[1] for v in agg.values())
ours = sum(v[1] for k, v in agg.items() if k.startswith("aft::"))
print(f"one training step: {len(win)} dispatches, span {span:.0f} us, kernel time {busy:.0f} us "
      f"({ours:.0f} us = {100 * ours / busy:.0f} % in this library's kernels)")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {k:<72} calls={c:4d} total={t:9.1f} us  mean={t / c:8.2f} us")
if "--timeline" in sys.argv:   # every dispatch of the step in launch order: start offset, duration, gap to the previous kernel
    print("\ntimeline (us): start  duration  gap  kernel")
    prev_end = win[0][0]
    for s, e, n in win:
        n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))
        n = re.sub(r"^void ", "", n)[:60]
        print(f"  {(s - win[0][0]) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:6.1f}  {n}")
        prev_end = e
