#!/usr/bin/env python3
"""One B=128 forward per step against two concurrent B=64 forwards on two streams (same frames): does tail filling
across independent half-batches beat the single persistent-grid launch sequence?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adafortitran_amd import _abi, synth
from adafortitran_amd.hip_ops import engine_from_numpy
SPEC = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4)
HID = (7, 42, 560)
sd = synth.make_state_dict(**SPEC, adaptive_hidden=HID, seed=1)
cfg = _abi.make_config(**SPEC, adaptive_hidden=HID)
BT = int(os.environ.get("AFT_BATCH", "128"))   # total frames; the halves are BT / 2 each
inp = synth.make_inputs(BT, seed=2)
dev = lambda a: torch.from_numpy(a).cuda()
pil, meta = dev(inp["pilots"]), [dev(inp[k]) for k in ("snr", "ds", "dop")]
eng = engine_from_numpy(cfg, sd, "cuda:0")
engs = [engine_from_numpy(cfg, sd, "cuda:0") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
H = BT // 2
halves = [(pil[:H].contiguous(), [m[:H].contiguous() for m in meta]), (pil[H:].contiguous(), [m[H:].contiguous() for m in meta])]

def one():
    eng.forward(pil, *meta)

def two():
    for e, s, (p, m) in zip(engs, streams, halves):
        with torch.cuda.stream(s):
            e.forward(p, *m)

for name, fn in ((f"one B={BT}", one), (f"two B={BT // 2} streams", two), (f"one B={BT}", one), (f"two B={BT // 2} streams", two)):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"{name}: {dt * 1e3:.4f} ms per {BT} frames = {BT / dt:.0f} frames/s")
