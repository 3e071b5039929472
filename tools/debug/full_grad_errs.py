#!/usr/bin/env python3
"""The full-depth gradient fixtures against float64, per tensor, worst first (tests/test_train_golden.py::_check64's numbers)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_train_golden as T
name = sys.argv[1] if len(sys.argv) > 1 else "G_grad_forti_full"
g, model, loss = T._step(name, "cuda") if hasattr(T, "_step") else T._run(name, "cuda")
g64 = T.Golden(name.replace("G_grad_", "G_grad64_"))
grads = {n: p.grad.detach().reshape(-1).cpu().numpy() for n, p in model.named_parameters()}
import hashlib, socket
print("host", socket.gethostname(), "loss", repr(loss), "md5 of all gradients", hashlib.md5(np.concatenate([grads[n] for n in sorted(grads)]).tobytes()).hexdigest())
for n in ("transformer_encoder.linear_2.weight", "transformer_encoder.linear_1.weight", "final_refiner.conv_block.0.weight", "transformer_encoder.transformer.layers.5.linear2.weight"):
    print("   ", n, hashlib.md5(grads[n].tobytes()).hexdigest()[:10])
errs = T._errors64(g64, grads)
be, bn = T.BASE64[name]
rows = []
for n, (e, en) in errs.items():
    te, tn = be + 2 * float(g64[f"gcond__{n}"]), bn + 2 * float(g64[f"gcondnorm__{n}"])
    rows.append((max(e / te, en / tn), n, e, te, en, tn))
for r in sorted(rows, reverse=True)[:12]:
    print(f"{r[0]:5.2f} of tol  {r[1]:55s} elem {r[2]:.2e}/{r[3]:.2e}  norm {r[4]:.2e}/{r[5]:.2e}")
