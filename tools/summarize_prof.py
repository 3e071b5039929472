#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output directories: per-kernel mean duration (kernel trace) and
per-kernel mean counter values (--pmc passes).  Usage: summarize_prof.py DIR [DIR...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(chain_kernel<[^>]*>|chain_split_kernel<[^>]*>|chain_bwd_kernel<[^>]*>|encoder_plane_kernel<[^>]*>|attn_kernel<[^>]*>|attn_split_kernel|"
                  r"conv_stream16_kernel<[^>]*>|conv_stream_kernel<[^>]*>|conv_rows16_kernel<[^>]*>|conv_rows_kernel<[^>]*>|prologue_kernel|conv_stack_kernel<[^>]*>|conv_stack_kernel|embed_kernel<[^>]*>|adapter_kernel|mse_kernel|linear_kernel|pack_weights_kernel)", name)
    return m.group(1) if m else name.replace("(anonymous namespace)::", "")[:60]


for d in sys.argv[1:]:
    for path in sorted(glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)):
        base = os.path.basename(path)
        rows = list(csv.DictReader(open(path)))
        if not rows:
            continue
        if base.endswith("kernel_trace.csv"):
            agg = defaultdict(list)
            for r in rows:
                agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            print(f"== {path}")
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                print(f"  {k:40s} calls={len(v):4d} mean={sum(v)/len(v)/1e3:9.2f} us  min={min(v)/1e3:9.2f} us  total={sum(v)/1e6:8.3f} ms")
        elif base.endswith("counter_collection.csv"):
            agg = defaultdict(lambda: defaultdict(list))
            for r in rows:
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            print(f"== {path}")
            for k, cs in agg.items():
                print("  " + k)
                for cn, v in sorted(cs.items()):
                    print(f"      {cn:32s} mean={sum(v)/len(v):16.1f}  n={len(v)}")
