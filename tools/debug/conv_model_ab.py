#!/usr/bin/env python3
"""The reduced model step (2 layers, B = 16) of tests/test_train_golden.py with the banded and with the streaming conv training
kernel (AFT_CONV_BANDED), each in its own process: per-tensor max|g_stream - g_banded| / max|g_banded|, largest first."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import test_train_golden as t
    from adafortitran_amd import synth
    from helpers import DEFAULT_SPEC
    spec = dict(DEFAULT_SPEC, num_layers=2)
    sd = synth.make_state_dict(**spec, adaptive_hidden=None, seed=4321)
    inp = synth.make_inputs(16, seed=4322)
    g = t._fresh_step(spec, False, "cuda", torch.float32, inp, sd, hip=True)
    np.savez(sys.argv[2], **g)
    sys.exit(0)
res = {}
for label, env in (("banded", {"AFT_CONV_BANDED": "1"}), ("stream", {})):
    e = dict(os.environ); e.pop("AFT_CONV_BANDED", None); e.update(env)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", f"/tmp/ab_{label}.npz"], env=e, check=True)
    res[label] = dict(np.load(f"/tmp/ab_{label}.npz"))
rows = sorted(((np.abs(res["stream"][k] - res["banded"][k]).max() / np.abs(res["banded"][k]).max(), k) for k in res["banded"]), reverse=True)
for e, k in rows[:12]:
    print(f"{e:10.2e}  {k}")
