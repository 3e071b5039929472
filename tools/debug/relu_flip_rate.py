#!/usr/bin/env python3
"""tests/test_train_golden.py::test_hip_gradients_are_as_close_to_float64_as_pytorch_fp32's first stage, per seed: how far the
HIP path's conv gradient maps are from PyTorch-ROCm's (max deviation / map max, share of pixels above the test's threshold)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_train_golden as T
from adafortitran_amd import synth
from helpers import DEFAULT_SPEC
adaptive = len(sys.argv) > 1 and sys.argv[1] == "ada"
spec = dict(DEFAULT_SPEC, num_layers=2)
sd = synth.make_state_dict(**spec, adaptive_hidden=(7, 42, 560) if adaptive else None, seed=4321)
thr = 5e-4 if adaptive else 2e-6
for seed in range(4322, 4334):
    inp = synth.make_inputs(16, seed=seed)
    m_rocm, m_hip = {}, {}
    T._fresh_step(spec, adaptive, "cuda", torch.float32, inp, sd, hip=False, maps=m_rocm)
    T._fresh_step(spec, adaptive, "cuda", torch.float32, inp, sd, hip=True, maps=m_hip)
    out = []
    for name in ("final", "initial"):
        a, b = m_rocm[name][0], m_hip[name][0]
        d = np.abs(a - b) / np.abs(a).max()
        out.append(f"{name}: max {d.max():.1e} above {(d > thr).mean():.2%}")
    print(seed, " | ".join(out), flush=True)
