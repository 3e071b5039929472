#!/usr/bin/env python3
"""Host time of each phase of a training step at the bench batch, no synchronisation inside the loop: a phase whose host time tracks the
DEVICE time of earlier work contains a blocking call."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import train_bench  # noqa: E402
from adafortitran_amd import synth  # noqa: E402
from adafortitran_amd.optim import ShardedFlatAdam  # noqa: E402
B = int(os.environ.get("AFT_BATCH", "128"))
meta_on = os.environ.get("AFT_META", "cpu")
model = train_bench.build("adafortitran", 0.1).train()
opt = ShardedFlatAdam(model.parameters(), lr=1e-4)
inp = synth.make_inputs(B, seed=5)
pil = torch.from_numpy(inp["pilots"]).cuda(); meta = synth.meta_tuple(inp)
if meta_on == "cuda":
    meta = tuple(x.cuda() if torch.is_tensor(x) else x for x in meta)
tgt = torch.randn(B, 120, 14, dtype=torch.complex64, device="cuda")
names = ["zero_grad", "forward", "loss", "backward", "opt.step"]
acc = [0.0] * 5
def step(rec):
    t = [time.perf_counter()]
    opt.zero_grad(); t.append(time.perf_counter())
    out = model(pil, meta); t.append(time.perf_counter())
    loss = torch.nn.functional.mse_loss(torch.view_as_real(out), torch.view_as_real(tgt)); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if rec:
        for i in range(5): acc[i] += t[i + 1] - t[i]
for _ in range(5): step(False)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N): step(True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B} meta on {meta_on}: host loop {1e3 * (t1 - t0) / N:.3f} ms per step, drain after the loop {1e3 * (t2 - t1):.3f} ms, total {1e3 * (t2 - t0) / N:.3f} ms per step")
print("  " + "  ".join(f"{n} {1e3 * a / N:.3f}" for n, a in zip(names, acc)))
