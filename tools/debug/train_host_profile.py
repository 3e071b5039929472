#!/usr/bin/env python3
"""Where the HOST time of one HIP training step goes (cProfile over 30 steps at a small batch, where the GPU work hides nothing)."""
import cProfile, os, pstats, sys, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import train_bench  # noqa: E402
from adafortitran_amd import synth  # noqa: E402
from adafortitran_amd.optim import ShardedFlatAdam  # noqa: E402
B = int(os.environ.get("AFT_BATCH", "8"))
model = train_bench.build("adafortitran", 0.1).train()
opt = ShardedFlatAdam(model.parameters(), lr=1e-4)
inp = synth.make_inputs(B, seed=5)
pil = torch.from_numpy(inp["pilots"]).cuda(); meta = synth.meta_tuple(inp)
tgt = torch.randn(B, 120, 14, dtype=torch.complex64, device="cuda")
def step():
    opt.zero_grad()
    loss = torch.nn.functional.mse_loss(torch.view_as_real(model(pil, meta)), torch.view_as_real(tgt))
    loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(30): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
