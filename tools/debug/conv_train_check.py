#!/usr/bin/env python3
"""One ConvEnhancer stack (reference blocks/enhancers.py:12-20), forward + backward, five ways: float64 on the CPU (yardstick),
torch fp32 on the CPU, PyTorch-ROCm fp32 (MIOpen), this library's banded training kernel, this library's streaming training kernel.
Prints max|x - x64| / max|x64| for the output, the input gradient and every parameter gradient."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from adafortitran_amd import blocks

def run(device, dtype, hip, planes=32, seed=0):
    torch.manual_seed(seed)
    m = blocks.ConvEnhancer()
    x = torch.randn(planes, 1, 120, 14) * float(os.environ.get("SCALE_X", "1"))
    g = torch.randn(planes, 1, 120, 14) * float(os.environ.get("SCALE_G", "1"))
    if os.environ.get("SMOOTH"):      # a smooth field like an upsampled channel: neighbouring pixels nearly equal
        x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (3, 3, 3, 3), mode="replicate"), 7, stride=1)
    blocks.ConvEnhancer.hip_training = hip
    m = m.to(device=device, dtype=dtype)
    x = x.to(device=device, dtype=dtype).requires_grad_(True)
    y = m(x)
    y.backward(g.to(device=device, dtype=dtype))
    out = {"y": y, "dx": x.grad}
    for n, p in m.named_parameters():
        out["d" + n.replace("conv_block.", "c")] = p.grad
    return {k: v.detach().double().cpu().numpy() for k, v in out.items()}

if len(sys.argv) > 1 and sys.argv[1] == "child":
    r = run("cuda", torch.float32, True)
    np.savez(sys.argv[2], **r)
    sys.exit(0)
ref = run("cpu", torch.float64, False)
res = {"cpu fp32": run("cpu", torch.float32, False), "rocm fp32": run("cuda", torch.float32, False)}
for label, env in (("hip banded", {"AFT_CONV_BANDED": "1"}), ("hip stream", {})):
    path = f"/tmp/conv_check_{label.split()[1]}.npz"
    e = dict(os.environ, **env)
    e.pop("AFT_CONV_BANDED", None) if not env else None
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], env=e, check=True)
    res[label] = dict(np.load(path))
print(f"{'':12s}" + "".join(f"{k:>14s}" for k in res))
for k in ref:
    print(f"{k:12s}" + "".join(f"{np.abs(res[l][k] - ref[k]).max() / np.abs(ref[k]).max():14.2e}" for l in res))
