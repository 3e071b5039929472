import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from adafortitran_amd import _abi, synth
from adafortitran_amd.hip_ops import engine_from_numpy
SPEC = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4)
hid=(7,42,560)
sd = synth.make_state_dict(**SPEC, adaptive_hidden=hid, seed=20251114)
cfg = _abi.make_config(**SPEC, adaptive_hidden=hid)
eng = engine_from_numpy(cfg, sd, "cuda:0")
inp = synth.make_inputs(128, seed=20251114)
t=lambda a: torch.from_numpy(a).to("cuda:0")
meta=[t(inp[k]) for k in ("snr","ds","dop")]; pil=t(inp["pilots"])
full=eng.forward(pil,*meta).clone()
again=eng.forward(pil,*meta).clone()
print("determinism max diff", (torch.view_as_real(full)-torch.view_as_real(again)).abs().max().item())
for lo in (0,16,48,112):
    part=eng.forward(pil[lo:lo+16], *[m[lo:lo+16] for m in meta])
    d=(torch.view_as_real(part)-torch.view_as_real(full[lo:lo+16])).abs()
    print(lo, "max diff", d.max().item(), "n diff", (d>0).sum().item(), "per-frame", [round(x,9) for x in d.amax(dim=(1,2,3)).tolist()][:16])
