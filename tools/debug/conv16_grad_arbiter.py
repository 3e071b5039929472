"""Round 6: G_grad_forti_h24 with the 16x16x4 training conv kernel vs the 32x32x2 one (switch AFT_CONV_MFMA32): per tensor upstream of the
first conv stack's ReLUs, the error against the reference's fp32 fixture and against its FLOAT64 twin (who is right when they differ)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
from adafortitran_amd import _lib
from helpers import Golden
import test_train_golden as T

for name in sys.argv[1:] or ["G_grad_forti_h24"]:
    g64 = Golden(name.replace("G_grad_", "G_grad64_"))
    for sw in (None, "1"):
        _lib.set_switch("AFT_CONV_MFMA32", sw)
        g, model, loss = T._step(name, "cuda")
        worst = []
        for n, p in model.named_parameters():
            got = p.grad.detach().reshape(-1).cpu().numpy()[::T.STRIDE][:T.MAXN]
            e32 = float(np.abs(got - g[f"gsample__{n}"]).max() / float(g[f"gmax__{n}"]))
            e64 = float(np.abs(got.astype(np.float64) - g64[f"gsample__{n}"]).max() / float(g64[f"gmax__{n}"]))
            f64 = float(np.abs(g[f"gsample__{n}"].astype(np.float64) - g64[f"gsample__{n}"]).max() / float(g64[f"gmax__{n}"]))
            worst.append((e32, e64, f64, n))
        worst.sort(reverse=True)
        print(name, "kernel", "32x32x2" if sw else "16x16x4")
        for e32, e64, f64, n in worst[:6]:
            print(f"   {n:55s} hip-vs-fp32fixture {e32:.2e}  hip-vs-fp64 {e64:.2e}  fp32fixture-vs-fp64 {f64:.2e}")
    _lib.set_switch("AFT_CONV_MFMA32", None)
