#!/usr/bin/env python3
"""aft_forward_f32 with AFT_LANES = 1 .. 4 (the forward run as that many concurrent shares of the batch, aft_api.hip) at several batches:
rate, and the bits against the unsplit forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adafortitran_amd import _abi, synth
from adafortitran_amd import _lib   # switches change through the ABI (the library reads the environment once, at load)
from adafortitran_amd.hip_ops import engine_from_numpy
SPEC = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4)
HID = (7, 42, 560)
sd = synth.make_state_dict(**SPEC, adaptive_hidden=HID, seed=1)
cfg = _abi.make_config(**SPEC, adaptive_hidden=HID)
dev = lambda a: torch.from_numpy(a).cuda()
eng = engine_from_numpy(cfg, sd, "cuda:0")
for BT in [int(x) for x in os.environ.get("AFT_BATCHES", "16,32,64,96,128,256").split(",")]:
    inp = synth.make_inputs(BT, seed=2)
    pil, meta = dev(inp["pilots"]), [dev(inp[k]) for k in ("snr", "ds", "dop")]
    ref = None
    for L in ("1", "2", "3", "4", None):
        if L is None: _lib.set_switch("AFT_LANES", None)
        else: _lib.set_switch("AFT_LANES", L)
        for _ in range(10): out = eng.forward(pil, *meta)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(100): out = eng.forward(pil, *meta)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 100)
        if ref is None: ref = out.clone()
        print(f"B={BT} lanes={L or 'auto'}: {best * 1e3:.4f} ms = {BT / best:.0f} frames/s  same bits: {bool(torch.equal(torch.view_as_real(out), torch.view_as_real(ref)))}")
