#!/usr/bin/env python3
"""Quick per-kernel timing on the GPU box (events on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adafortitran_amd import _abi, synth
from adafortitran_amd.hip_ops import engine_from_numpy, profile_kernel
SPEC = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4)
HID = (7, 42, 560)
B = int(os.environ.get("AFT_BATCH", "128"))
if os.environ.get("AFT_CONFIG") == "C5":   # BASELINE config 5 per GPU: 240 x 28 grid, 12 layers, d = 256, 64 frames
    SPEC = dict(ofdm=(240, 28), pilot=(24, 4), patch=(3, 2), num_layers=12, model_dim=256, num_head=8)
    HID = (7, 42, 2240)
    B = int(os.environ.get("AFT_BATCH", "64"))
which = sys.argv[1:] or ["upsample", "embed", "qkv", "attention", "chain", "tail"]
tokens = (SPEC["ofdm"][0] // 3) * (SPEC["ofdm"][1] // 2)
sd = synth.make_state_dict(**SPEC, adaptive_hidden=HID, max_seq_len=max(512, tokens), seed=20251114)
cfg = _abi.make_config(**SPEC, adaptive_hidden=HID)
eng = engine_from_numpy(cfg, sd, "cuda:0")
inp = synth.make_inputs(B, ofdm=SPEC["ofdm"], pilot=SPEC["pilot"], seed=20251114)
dev = lambda a: torch.from_numpy(a).to("cuda:0")
pil, meta = dev(inp["pilots"]), [dev(inp[k]) for k in ("snr", "ds", "dop")]
out = torch.empty((B, *SPEC["ofdm"]), dtype=torch.complex64, device="cuda:0")
from adafortitran_amd import _lib
stamps = _lib.get_switch("AFT_STAMPS")   # only meaningful with a --diag build; off for the first forward
_lib.set_switch("AFT_STAMPS", None)
eng.forward(pil, *meta, out=out); torch.cuda.synchronize()
if stamps: _lib.set_switch("AFT_STAMPS", stamps)
res = {}
for name in which:
    io = pil if name == "upsample" else (out if name == "tail" else None)
    if name == "prologue":   # [pilots | snr | ds | dop] as one buffer (AFT_KERNEL_PROLOGUE)
        io = torch.cat([torch.view_as_real(pil).reshape(-1)] + [m.reshape(-1).float() for m in meta]).contiguous()
    profile_kernel(eng, name, B, 3, io); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); profile_kernel(eng, name, B, 20, io); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    res[name] = round(best * 1e3, 1)
print(res)
