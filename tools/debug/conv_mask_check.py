#!/usr/bin/env python3
"""ReLU patterns of final_refiner's three hidden stages in the reduced model step: float64 vs float32 on the CPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import adafortitran_amd as A
from adafortitran_amd import synth
from helpers import DEFAULT_SPEC
def run(dtype):
    spec = dict(DEFAULT_SPEC, num_layers=2)
    sd = synth.make_state_dict(**spec, adaptive_hidden=None, seed=4321)
    inp = synth.make_inputs(16, seed=4322)
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    mc = A.ModelConfig(model_type="fortitran", patch_size=(3, 2), num_layers=2, model_dim=128, num_head=4, max_seq_len=512, device="cpu", dropout=0.0)
    model = A.FortiTranEstimator(sc, mc)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    if dtype == torch.float64: model.double()
    model.train()
    acts = {}
    for i in (1, 3, 5):
        model.final_refiner.conv_block[i].register_forward_hook(lambda m, a, o, i=i: acts.setdefault(i, []).append(o.detach().double().numpy()))
    cdt = torch.complex128 if dtype == torch.float64 else torch.complex64
    out = model(torch.from_numpy(inp["pilots"]).to(cdt))
    return {i: np.concatenate(v) for i, v in acts.items()}
a64, a32 = run(torch.float64), run(torch.float32)
for i in (1, 3, 5):
    m64, m32 = a64[i] > 0, a32[i] > 0
    diff = m64 != m32
    print(f"stage {i}: {diff.sum()} of {diff.size} ReLU decisions differ; |act| at those: 64: {np.abs(a64[i][diff]).max() if diff.any() else 0:.2e}  32: {np.abs(a32[i][diff]).max() if diff.any() else 0:.2e}; "
          f"exact zeros 64: {(a64[i] == 0).mean():.3f}  max|act32 - act64| {np.abs(a32[i] - a64[i]).max():.2e}")
