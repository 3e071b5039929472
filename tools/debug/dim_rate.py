#!/usr/bin/env python3
"""Forward rate of the default grid at other model dims (AFT_DIMS="160:5,192:6", AFT_BATCHES="64,128"); run once per library build
(AFT_LIB_PATH) to A/B a build switch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adafortitran_amd import _abi, synth
from adafortitran_amd.hip_ops import engine_from_numpy
dev = lambda a: torch.from_numpy(a).cuda()
for item in os.environ.get("AFT_DIMS", "160:5,192:6").split(","):
    d, heads = (int(x) for x in item.split(":"))
    spec = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=d, num_head=heads)
    sd = synth.make_state_dict(**spec, adaptive_hidden=(7, 42, 560), seed=1)
    cfg = _abi.make_config(**spec, adaptive_hidden=(7, 42, 560))
    eng = engine_from_numpy(cfg, sd, "cuda:0")
    for B in [int(x) for x in os.environ.get("AFT_BATCHES", "64,128").split(",")]:
        inp = synth.make_inputs(B, seed=2)
        pil, meta = dev(inp["pilots"]), [dev(inp[k]) for k in ("snr", "ds", "dop")]
        for _ in range(20): out = eng.forward(pil, *meta)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(50): out = eng.forward(pil, *meta)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 50)
        print(f"d={d} heads={heads} B={B}: {best * 1e3:.4f} ms = {B / best:.0f} frames/s  checksum {float(torch.view_as_real(out).double().abs().sum()):.9e}")
