import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import adafortitran_amd as A
from helpers import Golden
from test_estimators_cpu import _configs, golden_meta
name = sys.argv[1] if len(sys.argv) > 1 else "H24_ada_d96_heads4"
eval_first = len(sys.argv) > 2 and sys.argv[2] == "eval"
g = Golden(name)
cls = A.AdaFortiTranEstimator if g.adaptive else A.FortiTranEstimator
pil = torch.from_numpy(g["pilots"]); meta = golden_meta(g) if g.adaptive else None
if eval_first:
    sc, mc = _configs(g.spec, device="cuda")
    model = cls(sc, mc); model.load_state_dict({k: torch.from_numpy(v) for k, v in g.state_dict().items()}); model.eval()
    with torch.no_grad():
        out = model(pil, meta) if g.adaptive else model(pil)
    print("eval err", np.abs(out.cpu().numpy() - g["out"]).max() / np.abs(g["out"]).max())
grads = []
for dev in ("cpu", "cuda"):
    s_, m_ = _configs(dict(g.spec, dropout=0.0), device=dev)
    mdl = cls(s_, m_)
    mdl.load_state_dict({k: torch.from_numpy(v) for k, v in g.state_dict().items()})
    mdl.train()
    est = mdl(pil, meta) if g.adaptive else mdl(pil)
    tgt = torch.from_numpy(g["target"]).to(dev)
    torch.view_as_real(est - tgt).pow(2).mean().backward()
    grads.append({n: p.grad.detach().cpu().numpy() for n, p in mdl.named_parameters()})
    print(dev, "out", est.detach().abs().max().item(), mdl.training_backends() if dev == "cuda" else "")
for n, ref in grads[0].items():
    e = np.abs(grads[1][n] - ref).max() / (np.abs(ref).max() + 1e-30)
    if e > 2e-4: print("%-70s %.2e  |g|max %.2e" % (n, e, np.abs(ref).max()))
# the same step in float64 and with the library's training kernels switched off
from adafortitran_amd import training
def run(dev, hip, dbl):
    s_, m_ = _configs(dict(g.spec, dropout=0.0), device=dev)
    mdl = cls(s_, m_)
    mdl.load_state_dict({k: torch.from_numpy(v) for k, v in g.state_dict().items()})
    mdl.train()
    mdl.transformer_encoder.hip_training = hip
    mdl.initial_enhancer.hip_training = mdl.final_refiner.hip_training = hip
    training.HipLinear.default_hip_training = hip
    if hasattr(mdl, "channel_adapter"): mdl.channel_adapter.hip_training = hip
    p, m, t = pil, meta, torch.from_numpy(g["target"]).to(dev)
    if dbl:
        mdl = mdl.double(); p = pil.to(torch.complex128); t = t.to(torch.complex128)
        m = tuple(x.double() if torch.is_tensor(x) and x.is_floating_point() else x for x in meta) if meta is not None else None
    est = mdl(p, m) if g.adaptive else mdl(p)
    torch.view_as_real(est - t).pow(2).mean().backward()
    training.HipLinear.default_hip_training = True
    return {n: q.grad.detach().cpu().numpy() for n, q in mdl.named_parameters()}
f64 = run("cuda", False, True)
for tag, gr in (("cpu(first loop)", grads[0]), ("cuda(first loop)", grads[1]), ("hip again", run("cuda", True, False)), ("rocm32", run("cuda", False, False))):
    for n in ("pilot_upsampler.weight", "initial_enhancer.conv_block.0.weight", "final_refiner.conv_block.0.weight"):
        print("%-18s %-40s vs f64 %.2e" % (tag, n, np.abs(gr[n] - f64[n]).max() / np.abs(f64[n]).max()))
