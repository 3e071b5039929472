#!/usr/bin/env python3
"""Per kernel: mean SQ_INSTS_MFMA, SQ_INSTS_VALU and their ratio from rocprofv3 --pmc CSV output.  Usage: pmc_ratio.py DIR"""
import collections
import csv
import glob
import sys

rows = []
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))


def mean(c, k):
    return sum(c[k]) / max(len(c[k]), 1) if k in c else 0.0


for n, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    mf, va = mean(c, "SQ_INSTS_MFMA"), mean(c, "SQ_INSTS_VALU")
    if mf > 0:
        busy = mean(c, "SQ_VALU_MFMA_BUSY_CYCLES")
        print(f"{n:64s} n={len(c['SQ_INSTS_MFMA']):3d}  MFMA {mf / 1e6:7.2f} M  VALU {va / 1e6:7.2f} M  VALU/MFMA {va / mf:5.2f}  "
              f"MFMA-busy cycles {busy / 1e6:8.1f} M")
