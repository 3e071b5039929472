#!/usr/bin/env python3
"""Fill the caching allocator's pool with NaN-poisoned blocks, then check that a B=16 forward reproduces the
matching slice of a B=128 forward bit for bit (what tests/test_hip_parity.py::test_full_size_batch128_properties
asserts): any kernel that lets uninitialised workspace reach its arithmetic shows up here."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from adafortitran_amd import _abi, synth
from adafortitran_amd.hip_ops import engine_from_numpy

poison = float(os.environ.get("POISON", "nan"))
blocks = [torch.full((n,), poison, device="cuda") for n in (1 << 26, 1 << 25, 1 << 24, 1 << 22, 1 << 20, 1 << 18) for _ in range(3)]
del blocks
SPEC = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4)
hid = (7, 42, 560)
sd = synth.make_state_dict(**SPEC, adaptive_hidden=hid, seed=20251114)
cfg = _abi.make_config(**SPEC, adaptive_hidden=hid)
eng = engine_from_numpy(cfg, sd, "cuda:0")
inp = synth.make_inputs(128, seed=20251114)
t = lambda a: torch.from_numpy(a).cuda()
meta = [t(inp[k]) for k in ("snr", "ds", "dop")]
pil = t(inp["pilots"])
full = eng.forward(pil, *meta).clone()
print("full finite:", bool(torch.isfinite(torch.view_as_real(full)).all()))
for lo in (0, 16, 48, 112):
    part = eng.forward(pil[lo:lo + 16], *[m[lo:lo + 16] for m in meta]).clone()
    d = (torch.view_as_real(part) - torch.view_as_real(full[lo:lo + 16])).abs()
    bad = torch.nonzero(d.amax(dim=(1, 2, 3)) > 0).flatten().tolist()
    print(f"lo={lo}: max|diff|={float(d.max()):.3e} finite={bool(torch.isfinite(torch.view_as_real(part)).all())} frames differing: {bad}")
    if bad:
        f = bad[0]
        w = torch.nonzero(d[f].amax(dim=2) > 0)
        print("   frame", f, "pixels differing:", len(w), "rows", sorted(set(w[:, 0].tolist()))[:20], "cols", sorted(set(w[:, 1].tolist())))
again = eng.forward(pil, *meta)
print("again == full:", bool(torch.equal(torch.view_as_real(again), torch.view_as_real(full))))
