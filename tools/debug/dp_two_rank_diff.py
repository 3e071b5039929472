#!/usr/bin/env python3
"""tests/test_rccl_gpu.py::test_two_ranks_sharing_the_gpu_train_like_one_process, taken apart: which parameter tensors differ between two
ranks x 4 frames (ShardedFlatAdam) and one process x 8 frames (torch Adam) after three steps, and by how much."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.multiprocessing as mp
import test_rccl_gpu as T

if __name__ == "__main__":
    port = T._free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(T._dp_worker, args=(2, port, ret), nprocs=2, join=True)
        got = dict(ret)
    want, want_grad = T._dp_train(0, 1)
    model, *_ = T._dp_model_and_data(8)
    dpar = np.abs(got[0][0] - want)
    dg = np.abs(got[0][1] - want_grad)
    print("q9999", np.quantile(dpar, 0.9999), "max", dpar.max(), "gerr/max", dg.max() / np.abs(want_grad).max())
    off = 0
    for name, p in model.named_parameters():
        n = p.numel()
        d, g, gd = dpar[off:off + n], np.abs(want_grad[off:off + n]), dg[off:off + n]
        print(f"{name:60s} n={n:7d} dpar max {d.max():.2e} >1.5e-4: {int((d > 1.5e-4).sum()):5d}  |g|max {g.max():.2e} med {np.median(g):.2e}  dg max {gd.max():.2e}")
        off += n
