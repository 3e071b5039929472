#!/usr/bin/env python3
"""Forwards of a configuration the packed engine does not take (aft_engine_of = GENERAL: model_dim 512, 8 heads, default grid, 128
frames) for a rocprofv3 kernel trace (tools/profile_round.sh -> profiles/rNN_general_kernel_trace_summary.txt)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench

c = dict(bench.C3, model_dim=int(os.environ.get("AFT_DIM", "512")), num_head=int(os.environ.get("AFT_HEADS", "8")), batch=int(os.environ.get("AFT_BATCH", "128")))
wl = bench.Workload(c, torch.device("cuda", 0))
for _ in range(int(os.environ.get("AFT_FWD", "10"))):
    wl.step()
torch.cuda.synchronize()
wall, _, _ = bench.timed_steps(wl, wl.step, 10, 0, torch.cuda.synchronize)
print({"config": f"d={c['model_dim']} heads={c['num_head']} B={wl.B}", "frames_per_s": round(wl.B * 10 / wall, 1)})
