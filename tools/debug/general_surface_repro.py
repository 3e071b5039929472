"""Round 6 repro: the module surface (CPU inputs through the pinned ring) on a general-engine FortiTran against the engine on device inputs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import adafortitran_amd as A
from helpers import Golden
from test_estimators_cpu import _configs

def train_step(name):
    g = Golden(name)
    from test_estimators_cpu import golden_meta
    sc_g, mc_g = _configs(dict(g.spec, dropout=0.0), device="cuda")
    cls = A.AdaFortiTranEstimator if g.adaptive else A.FortiTranEstimator
    mdl = cls(sc_g, mc_g)
    mdl.load_state_dict({k: torch.from_numpy(v) for k, v in g.state_dict().items()})
    mdl.train()
    pil = torch.from_numpy(g["pilots"]); meta = golden_meta(g) if g.adaptive else None
    est = mdl(pil, meta) if g.adaptive else mdl(pil)
    tgt = torch.from_numpy(g["target"]).cuda()
    torch.view_as_real(est - tgt).pow(2).mean().backward()
    torch.cuda.synchronize()
    print("trained one step of", name)

names = [a for a in sys.argv[1:] if not a.startswith("+")]
for pre in [a[1:] for a in sys.argv[1:] if a.startswith("+")]:
    train_step(pre)
for name in names or ["H128_forti_d256_heads2"]:
    g = Golden(name)
    sc, mc = _configs(g.spec, device="cuda")
    cls = A.AdaFortiTranEstimator if g.adaptive else A.FortiTranEstimator
    model = cls(sc, mc)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g.state_dict().items()})
    model.eval()
    pil = torch.from_numpy(g["pilots"])
    from test_estimators_cpu import golden_meta
    meta = golden_meta(g) if g.adaptive else None
    with torch.no_grad():
        out = model(pil, meta) if g.adaptive else model(pil)
        out_dev = model(pil.cuda(), meta) if g.adaptive else model(pil.cuda())
    eng = model._engine
    ref = g["out"]
    print(name, "engine", model.hip_engine_name(), "cpu-in err", float(np.abs(out.cpu().numpy() - ref).max()), "dev-in err",
          float(np.abs(out_dev.cpu().numpy() - ref).max()), "|ref|max", float(np.abs(ref).max()))
    for reg in ("conv_enhanced", "enc_out"):
        r = eng.forward_region(reg, pil.shape[0]).cpu().numpy()
        want = g[reg] if reg in g else None
        print("  region", reg, r.shape, "absmax", float(np.abs(r).max()), "finite", bool(np.isfinite(r).all()),
              "" if want is None else ("err %g" % float(np.abs(r.reshape(want.shape[0], -1)[:, :1] - want.reshape(want.shape[0], -1)[:, :1]).max())))
    # second call (warm)
    with torch.no_grad():
        out2 = model(pil)  if not g.adaptive else model(pil, meta)
    print("  second call err", float(np.abs(out2.cpu().numpy() - ref).max()))
