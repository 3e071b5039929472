"""Round 6: a seed for the G_grad_forti_h24 fixture whose step has no ReLU decision at fp32 rounding: the HIP step with the 16x16x4 and with
the 32x32x2 training conv kernel, and PyTorch-ROCm autograd fp32, all against the same module in float64 on the GPU."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import adafortitran_amd as A
from adafortitran_amd import _lib, synth

spec = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=2, model_dim=96, num_head=4)
def step(seed, mode):
    sd = synth.make_state_dict(**spec, adaptive_hidden=None, max_seq_len=512, seed=seed, attn_gain=8.0)
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    kw = dict(model_type="fortitran", patch_size=(3, 2), num_layers=2, model_dim=96, num_head=4, max_seq_len=512, device="cuda", dropout=0.0)
    model = A.FortiTranEstimator(sc, A.ModelConfig(**kw))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.train()
    inp = synth.make_inputs(3, seed=seed + 1)
    pil, tgt = torch.from_numpy(inp["pilots"]), torch.from_numpy(inp["target"]).cuda()
    if mode == "f64":
        model.double(); pil = pil.to(torch.complex128); tgt = tgt.to(torch.complex128)
    _lib.set_switch("AFT_CONV_MFMA32", "1" if mode == "hip32" else None)
    if mode == "torch":
        for m in model.modules():
            if hasattr(m, "hip_training"): m.hip_training = False
    out = model(pil)
    cat = lambda z: torch.cat((torch.real(z), torch.imag(z)), dim=1)
    torch.nn.MSELoss()(cat(out), cat(tgt)).backward()
    _lib.set_switch("AFT_CONV_MFMA32", None)
    return {n: p.grad.detach().double().cpu().numpy().ravel() for n, p in model.named_parameters()}
for seed in [784] + list(range(7840, 7860)):
    ref = step(seed, "f64")
    worst = {}
    for mode in ("hip16", "hip32", "torch"):
        g = step(seed, mode)
        worst[mode] = max(float(np.abs(g[n] - ref[n]).max() / np.abs(ref[n]).max()) for n in ref)
    print(seed, {k: f"{v:.1e}" for k, v in worst.items()}, "CLEAN" if max(worst.values()) < 5e-5 else "", flush=True)
