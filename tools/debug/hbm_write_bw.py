#!/usr/bin/env python3
"""HBM write and copy bandwidth as PyTorch's own kernels see it (fill_, copy_) -- the bound of the training kernels' tape stores."""
import torch

n = 1 << 28   # 1 GiB of floats
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")


def timed(f, reps=10):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


t = timed(lambda: x.fill_(1.0))
print(f"fill  1 GiB: {t * 1e6:8.1f} us  write {4 * n / t / 1e12:.2f} TB/s")
t = timed(lambda: y.copy_(x))
print(f"copy  1 GiB: {t * 1e6:8.1f} us  read+write {8 * n / t / 1e12:.2f} TB/s (write {4 * n / t / 1e12:.2f})")
t = timed(lambda: x.sum())
print(f"sum   1 GiB: {t * 1e6:8.1f} us  read {4 * n / t / 1e12:.2f} TB/s")
for mb in (36, 110, 294):
    m = mb * (1 << 20) // 4
    t = timed(lambda: x[:m].fill_(2.0), reps=50)
    print(f"fill {mb:4d} MB: {t * 1e6:8.1f} us  write {4 * m / t / 1e12:.2f} TB/s")
