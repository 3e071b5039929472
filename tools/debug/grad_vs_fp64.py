#!/usr/bin/env python3
"""Which part of the HIP training path is how far from the TRUE gradient?  One train_epoch-style step (dropout 0) of the
full-depth default model at B = 128 on the GPU, with the hand-written kernels switched on per block, every parameter's
gradient compared with the float64 run of the reference (tests/golden/G_grad64_*_full.npz) -- next to the reference's own
fp32 gradient (G_grad_*_full.npz) against the same yardstick.

    python tools/debug/grad_vs_fp64.py [forti|ada] [--small]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import adafortitran_amd as A
from adafortitran_amd import blocks, synth, training
from helpers import Golden

which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "forti"
suffix = "" if "--small" in sys.argv else "_full"
g32, g64 = Golden(f"G_grad_{which}{suffix}"), Golden(f"G_grad64_{which}{suffix}")
names = [str(n) for n in g64["names"]]


def errors(sample_of):
    out = {}
    for n in names:
        gm = float(g64[f"gmax__{n}"])
        out[n] = float(np.abs(sample_of(n).astype(np.float64) - g64[f"gsample__{n}"]).max() / gm)
    return out


def step(enc, conv, lin, ada):
    blocks.TransformerEncoderForChannels.hip_training = enc
    blocks.ConvEnhancer.hip_training = conv
    blocks.ChannelAdapter.hip_training = ada
    training.HipLinear.default_hip_training = lin
    g = g32
    s = g.spec
    sc = A.SystemConfig(ofdm=dict(num_scs=s["ofdm"][0], num_symbols=s["ofdm"][1]), pilot=dict(num_scs=s["pilot"][0], num_symbols=s["pilot"][1]))
    kw = dict(model_type="adafortitran" if g.adaptive else "fortitran", patch_size=tuple(s["patch"]), num_layers=s["num_layers"],
              model_dim=s["model_dim"], num_head=s["num_head"], activation=s.get("activation", "gelu"), max_seq_len=512,
              pos_encoding_type="learnable", device="cuda", dropout=s["dropout"])
    if g.adaptive:
        kw.update(channel_adaptivity_hidden_sizes=list(s["adaptive_hidden"]), adaptive_token_length=6)
    model = (A.AdaFortiTranEstimator if g.adaptive else A.FortiTranEstimator)(sc, A.ModelConfig(**kw))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g.state_dict().items()}, strict=True)
    model.train()
    if "pilots" in g:
        inp = {k: g[k] for k in ("pilots", "target")}
        if g.adaptive:
            inp.update({k: g[k] for k in ("snr", "ds", "dop")})
    else:
        inp = synth.make_inputs(g.meta["batch"], ofdm=tuple(s["ofdm"]), pilot=tuple(s["pilot"]), seed=s["seed"] + 1)
    pil, tgt = torch.from_numpy(inp["pilots"]), torch.from_numpy(inp["target"]).cuda()
    meta = synth.meta_tuple(inp) if g.adaptive else None
    out = model(pil, meta) if meta is not None else model(pil)
    cat = lambda z: torch.cat((torch.real(z), torch.imag(z)), dim=1)  # noqa: E731
    loss = torch.nn.MSELoss()(cat(out), cat(tgt))
    loss.backward()
    params = dict(model.named_parameters())
    return {n: params[n].grad.detach().reshape(-1).cpu().numpy()[::7][:4096] for n in names}


ref = errors(lambda n: g32[f"gsample__{n}"])
cases = [("torch-rocm", (False, False, False, False)), ("hip all", (True, True, True, True)), ("hip encoder only", (True, False, False, False)),
         ("hip conv only", (False, True, False, False)), ("hip linear only", (False, False, True, False))]
if which == "ada":
    cases.append(("hip adapter only", (False, False, False, True)))
res = {}
for label, sw in cases:
    smp = step(*sw)
    res[label] = errors(lambda n: smp[n])
worst = sorted(names, key=lambda n: -res["hip all"][n])[:14]
print(f"{which}{suffix}: max over tensors of max|g - g64| / |g64|max")
print(f"  reference fp32 (CPU): {max(ref.values()):.2e}")
for label, _ in cases:
    print(f"  {label:18s}: {max(res[label].values()):.2e}   (ratio to reference, worst tensor: "
          f"{max(res[label][n] / max(ref[n], 1e-9) for n in names):.1f})")
print("worst tensors of the HIP path:")
for n in worst:
    print(f"  {n:62s} gmax {float(g64['gmax__' + n]):.1e} ref {ref[n]:.1e} | " + " ".join(f"{res[l][n]:.1e}" for l, _ in cases))
