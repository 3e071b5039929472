#!/usr/bin/env python3
"""Fused row-local backward chain (k_chain_bwd.hip) against the round-2 launch sequence on the same layer / inputs:
max relative difference of dx and of every parameter gradient, with and without dropout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adafortitran_amd import _abi
from adafortitran_amd import _lib   # switches change through the ABI (the library reads the environment once, at load)
from adafortitran_amd.training import HipEncoderLayerFunction, layer_params

def run(layer, cfg, x0, gout, p, seed, unfused):
    if unfused: _lib.set_switch("AFT_TRAIN_UNFUSED_BWD", "1")
    else: _lib.set_switch("AFT_TRAIN_UNFUSED_BWD", None)
    layer.zero_grad()
    x = x0.clone().requires_grad_(True)
    out = HipEncoderLayerFunction.apply(x, cfg, p, seed, *layer_params(layer))
    out.backward(gout)
    return [x.grad.clone()] + [q.grad.clone() for q in layer_params(layer)]

names = ["dx"] + list(_abi.LAYER_PARAM_NAMES)
for grid, planes in [((120, 14), int(x)) for x in os.environ.get("AFT_CHECK_PLANES", "2,16").split(",")]:
    for p in (0.0, 0.1):
        d, heads = 128, 4
        cfg = _abi.make_config(ofdm=grid, pilot=(12, 2), patch=(3, 2), num_layers=1, model_dim=d, num_head=heads)
        torch.manual_seed(0)
        layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=heads, dim_feedforward=2 * d, activation="gelu", dropout=p, batch_first=True).cuda().train()
        x0 = torch.randn(planes, cfg.tokens, d, device="cuda"); gout = torch.randn_like(x0)
        a = run(layer, cfg, x0, gout, p, 5, False); b = run(layer, cfg, x0, gout, p, 5, True)
        worst = max((float((u - v).abs().max() / (v.abs().max() + 1e-30)), n) for u, v, n in zip(a, b, names))
        print(f"grid {grid} planes {planes} p={p}: worst rel diff {worst[0]:.3e} ({worst[1]})", flush=True)
        if worst[0] > 1e-3:
            for u, v, n in zip(a, b, names):
                print(f"    {n:32s} {float((u - v).abs().max() / (v.abs().max() + 1e-30)):.3e}")
