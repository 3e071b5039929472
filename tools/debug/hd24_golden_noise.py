"""The H24 / H48 golden sets' training step: HIP kernels, PyTorch-ROCm fp32 and the CPU composite against the same module in float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import adafortitran_amd as A
from adafortitran_amd import training
from helpers import Golden
from test_estimators_cpu import _configs, golden_meta

def rel(a, b): return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())

for name in sys.argv[1:] or ["H24_ada_d96_heads4", "H16_ada_heads8", "H48_forti_d192_heads4"]:
    g = Golden(name)
    cls = A.AdaFortiTranEstimator if g.adaptive else A.FortiTranEstimator
    pil = torch.from_numpy(g["pilots"]); meta = golden_meta(g) if g.adaptive else None
    res = {}
    for tag, dev, hip, dbl in (("cpu", "cpu", False, False), ("hip", "cuda", True, False), ("rocm", "cuda", False, False), ("f64", "cuda", False, True)):
        sc, mc = _configs(dict(g.spec, dropout=0.0), device=dev)
        mdl = cls(sc, mc)
        mdl.load_state_dict({k: torch.from_numpy(v) for k, v in g.state_dict().items()})
        mdl.train()
        mdl.transformer_encoder.hip_training = hip
        mdl.initial_enhancer.hip_training = mdl.final_refiner.hip_training = hip
        training.HipLinear.default_hip_training = hip
        if hasattr(mdl, "channel_adapter"): mdl.channel_adapter.hip_training = hip
        p, m = pil, meta
        if dbl:
            mdl = mdl.double(); p = pil.to(torch.complex128)
            m = tuple(x.double() if torch.is_tensor(x) and x.is_floating_point() else x for x in meta) if meta is not None else None
        est = mdl(p, m) if g.adaptive else mdl(p)
        tgt = torch.from_numpy(g["target"]).to(dev)
        if dbl: tgt = tgt.to(torch.complex128)
        torch.view_as_real(est - tgt).pow(2).mean().backward()
        res[tag] = {n: q.grad.detach().cpu().numpy() for n, q in mdl.named_parameters()}
    training.HipLinear.default_hip_training = True
    rows = sorted(((rel(res["hip"][n], res["f64"][n]), rel(res["rocm"][n], res["f64"][n]), rel(res["cpu"][n], res["f64"][n]), n) for n in res["f64"]
                   if not n.startswith("channel_adapter")), reverse=True)[:4]
    print(name)
    for r in rows: print("   vs f64: hip %.2e  rocm32 %.2e  cpu32 %.2e  %s" % r)
