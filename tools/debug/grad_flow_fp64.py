#!/usr/bin/env python3
"""Where does the fp32 backward on the GPU leave the true gradient?  The same train_epoch-style step (FortiTran default, 6 layers,
B = 128, dropout 0) three times: float64 on the CPU (the yardstick), float32 PyTorch-ROCm composite on the GPU, float32 with the
hand-written training kernels.  Module backward hooks capture the gradient ARRIVING at each stage's output (in backward order:
final_refiner, encoder as a whole, each encoder layer (PyTorch-ROCm run only), initial_enhancer, pilot_upsampler) and each is
compared with the float64 one: max|d| / max|g64| and ||d|| / ||g64||.

    python tools/debug/grad_flow_fp64.py [batch] [layers]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import adafortitran_amd as A
from adafortitran_amd import blocks, synth, training

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = int(sys.argv[2]) if len(sys.argv) > 2 else 6
SPEC = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=L, model_dim=128, num_head=4)
sd = synth.make_state_dict(**SPEC, adaptive_hidden=None, seed=779)
inp = synth.make_inputs(B, seed=780)


def run(device, dtype, hip):
    for cls in (blocks.TransformerEncoderForChannels, blocks.ConvEnhancer, blocks.ChannelAdapter):
        cls.hip_training = hip
    training.HipLinear.default_hip_training = hip
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    mc = A.ModelConfig(model_type="fortitran", patch_size=(3, 2), num_layers=L, model_dim=128, num_head=4, max_seq_len=512, device=device,
                       dropout=0.0)
    model = A.FortiTranEstimator(sc, mc)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    if dtype == torch.float64:
        model.double()
    model.train()
    grads = {}

    def hook(name):
        def fn(_m, gin, gout):
            grads["d_out " + name] = gout[0].detach().double().cpu().numpy()
        return fn

    mods = [("final_refiner", model.final_refiner), ("transformer_encoder", model.transformer_encoder),
            ("initial_enhancer", model.initial_enhancer), ("pilot_upsampler", model.pilot_upsampler)]
    mods += [(f"layer{i}", l) for i, l in enumerate(model.transformer_encoder.transformer.layers)]
    for name, m in mods:
        m.register_full_backward_hook(hook(name))
    cdt = torch.complex128 if dtype == torch.float64 else torch.complex64
    pil, tgt = torch.from_numpy(inp["pilots"]).to(cdt), torch.from_numpy(inp["target"]).to(cdt).to(device)
    out = model(pil)
    cat = lambda z: torch.cat((torch.real(z), torch.imag(z)), dim=1)  # noqa: E731
    loss = torch.nn.MSELoss()(cat(out), cat(tgt))
    loss.backward()
    for n, p in model.named_parameters():
        grads["param " + n] = p.grad.detach().double().cpu().numpy()
    return grads


g64 = run("cpu", torch.float64, False)
g32c = run("cpu", torch.float32, False)
runs = {"cpu fp32": g32c, "rocm fp32": run("cuda", torch.float32, False), "hip fp32": run("cuda", torch.float32, True)}
keys = [k for k in g64 if k.startswith("d_out")] + [k for k in g64 if k.startswith("param") and ("pilot_up" in k or "position" in k or "linear_" in k
                                                                                                  or "conv_block.0" in k or "layers.0." in k or "layers.5." in k)]
print(f"B = {B}, {L} layers: max|g - g64| / max|g64|   (||g - g64|| / ||g64||)")
print(f"{'':58s}" + "".join(f"{k:>24s}" for k in runs))
for k in keys:
    ref = g64[k]
    row = ""
    for name, g in runs.items():
        if k not in g or g[k].shape != ref.shape:
            row += f"{'-':>24s}"
            continue
        d = g[k] - ref
        row += f"{np.abs(d).max() / np.abs(ref).max():12.1e} ({np.linalg.norm(d) / np.linalg.norm(ref):8.1e})"
    print(f"{k:58s}{row}   |g64|max {np.abs(ref).max():.1e}")
