#!/usr/bin/env python3
"""Gradient arriving at / leaving the two conv stacks inside the reduced model step, five ways (float64 CPU = yardstick)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import adafortitran_amd as A
from adafortitran_amd import blocks, synth, training
from helpers import DEFAULT_SPEC

def run(device, dtype, hip):
    spec = dict(DEFAULT_SPEC, num_layers=2)
    sd = synth.make_state_dict(**spec, adaptive_hidden=None, seed=4321)
    inp = synth.make_inputs(16, seed=4322)
    for cls in (blocks.TransformerEncoderForChannels, blocks.ConvEnhancer, blocks.ChannelAdapter):
        cls.hip_training = hip
    training.HipLinear.default_hip_training = hip
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    mc = A.ModelConfig(model_type="fortitran", patch_size=(3, 2), num_layers=2, model_dim=128, num_head=4, max_seq_len=512, device=device, dropout=0.0)
    model = A.FortiTranEstimator(sc, mc)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    if dtype == torch.float64: model.double()
    model.train()
    cap = {}
    def hook(name):
        def fn(_m, gin, gout):
            key_o, key_i = name + ".gout", name + ".gin"
            go = gout[0].detach().double().cpu().numpy(); gi = gin[0].detach().double().cpu().numpy() if gin[0] is not None else None
            # CPU runs call each stack twice (Re pass, Im pass): stack them like the GPU's 2B planes ([Re planes; Im planes])
            cap.setdefault(key_o, []).append(go)
            if gi is not None: cap.setdefault(key_i, []).append(gi)
        return fn
    model.final_refiner.register_full_backward_hook(hook("final"))
    model.initial_enhancer.register_full_backward_hook(hook("initial"))
    cdt = torch.complex128 if dtype == torch.float64 else torch.complex64
    pil, tgt = torch.from_numpy(inp["pilots"]).to(cdt), torch.from_numpy(inp["target"]).to(cdt).to(device)
    out = model(pil)
    cat = lambda z: torch.cat((torch.real(z), torch.imag(z)), dim=1)
    torch.nn.MSELoss()(cat(out), cat(tgt)).backward()
    res = {}
    for k, v in cap.items():
        # backward order of the two CPU passes: Im first then Re (reverse of forward) -> [Re; Im]
        res[k] = np.concatenate(v[::-1], axis=0) if len(v) == 2 else v[0]
    res["y"] = out.detach().cpu().numpy().astype(np.complex128)
    return res

if len(sys.argv) > 1 and sys.argv[1] == "child":
    r = run("cuda", torch.float32, True)
    np.savez(sys.argv[2], **r); sys.exit(0)
ref = run("cpu", torch.float64, False)
res = {"cpu fp32": run("cpu", torch.float32, False), "rocm fp32": run("cuda", torch.float32, False)}
for label, env in (("hip banded", {"AFT_CONV_BANDED": "1"}), ("hip stream", {})):
    e = dict(os.environ); e.pop("AFT_CONV_BANDED", None); e.update(env)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", f"/tmp/ab2_{label.split()[1]}.npz"], env=e, check=True)
    res[label] = dict(np.load(f"/tmp/ab2_{label.split()[1]}.npz"))
print(f"{'':16s}" + "".join(f"{k:>14s}" for k in res) + "      |ref|max   shape")
for k in ref:
    row = ""
    for l in res:
        a = res[l].get(k)
        row += f"{np.abs(a - ref[k]).max() / np.abs(ref[k]).max():14.2e}" if a is not None and a.shape == ref[k].shape else f"{'shape':>14s}"
    print(f"{k:16s}{row}   {np.abs(ref[k]).max():10.2e}   {ref[k].shape}")
for k in ("final.gin",):
    r = ref[k]; c = res["cpu fp32"][k]; s_ = res["hip stream"][k]; b = res["hip banded"][k]
    dc, ds = c - r, s_ - r
    idx = np.unravel_index(np.abs(dc).argmax(), dc.shape)
    print(k, "max dev at", idx, "ref %.6e cpu %.6e stream %.6e banded %.6e" % (r[idx], c[idx], s_[idx], b[idx]))
    thr = 1e-4 * np.abs(r).max()
    print("  pixels with |dev| > 1e-4 max: cpu %d stream %d banded %d of %d; max|cpu - stream| / max|ref| = %.2e" %
          ((np.abs(dc) > thr).sum(), (np.abs(ds) > thr).sum(), (np.abs(b - r) > thr).sum(), r.size, np.abs(c - s_).max() / np.abs(r).max()))
    bad = np.argwhere(np.abs(dc) > thr)
    print("  planes:", sorted(set(bad[:, 0].tolist()))[:20], "rows:", sorted(set(bad[:, 2].tolist()))[:40], "cols:", sorted(set(bad[:, 3].tolist())))
