#!/usr/bin/env python3
"""Workload for rocprofv3: one full forward (B=128 AdaFortiTran default) then each kernel class
a few times through aft_profile_kernel_f32, so --kernel-trace/--pmc rows exist per kernel.
Usage (GPU box):  rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/prof_kernels.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adafortitran_amd import _abi, synth  # noqa: E402
from adafortitran_amd.hip_ops import engine_from_numpy, profile_kernel  # noqa: E402

SPEC = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4)
HID = (7, 42, 560)
B = int(os.environ.get("AFT_BATCH", "128"))
REPS = int(os.environ.get("AFT_REPS", "5"))
sd = synth.make_state_dict(**SPEC, adaptive_hidden=HID, seed=20251114)
cfg = _abi.make_config(**SPEC, adaptive_hidden=HID)
if os.environ.get("AFT_PRECISION") == "bf16x3":      # the opt-in split-precision tier (reported separately)
    cfg.precision = _abi.AFT_PRECISION_BF16X3
eng = engine_from_numpy(cfg, sd, "cuda:0")
inp = synth.make_inputs(B, seed=20251114)
dev = lambda a: torch.from_numpy(a).to("cuda:0")  # noqa: E731
pil, meta = dev(inp["pilots"]), [dev(inp[k]) for k in ("snr", "ds", "dop")]
out = torch.empty((B, 120, 14), dtype=torch.complex64, device="cuda:0")
for _ in range(int(os.environ.get("AFT_FWD", "3"))):
    eng.forward(pil, *meta, out=out)
torch.cuda.synchronize()
only = os.environ.get("AFT_ONLY")
for name, io in (("upsample", pil), ("embed", None), ("qkv", None), ("attention", None), ("chain", None), ("chain_last", None),
                 ("tail", out), ("encoder_plane", None)):
    if name == "encoder_plane" and not only:
        continue
    if only and name not in only.split(","):
        continue
    profile_kernel(eng, name, B, REPS, io)
    torch.cuda.synchronize()
print("done")
