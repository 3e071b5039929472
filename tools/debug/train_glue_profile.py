#!/usr/bin/env python3
"""Which PyTorch operators of one HIP training step launch the non-library kernels (fills, elementwise, reductions)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import train_bench  # noqa: E402
from adafortitran_amd import synth  # noqa: E402
from adafortitran_amd.optim import ShardedFlatAdam  # noqa: E402

model = train_bench.build("adafortitran", 0.1).train()
opt = ShardedFlatAdam(model.parameters(), lr=1e-4)
inp = synth.make_inputs(128, seed=5)
pil = torch.from_numpy(inp["pilots"]).cuda()
meta = synth.meta_tuple(inp)
tgt = torch.randn(128, 120, 14, dtype=torch.complex64, device="cuda")


def step():
    opt.zero_grad()
    loss = torch.nn.functional.mse_loss(torch.view_as_real(model(pil, meta)), torch.view_as_real(tgt))
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_time_total > 0 or e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:45]:
    print(f"{e.key[:70]:70s} calls={e.count:4d} self_dev={e.self_device_time_total:9.1f} us")
