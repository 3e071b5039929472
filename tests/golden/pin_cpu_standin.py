#!/usr/bin/env python3
"""Pin the CPU stand-in: bench.py's ``cpu_baseline`` times this package's estimator on
``device="cpu"`` because the reference cannot travel to the GPU box.  This script -- build container only, it imports
/root/reference -- runs BOTH on identical weights and inputs at the headline size (AdaFortiTran default, B = 128),
interleaved and with alternating order, and writes what it found to tests/golden/cpu_standin.json:

    max|out_standin - out_reference|   (must be 0: same torch.nn modules => same ATen / oneDNN / MKL kernels)
    median forward time of each, their ratio (must be within +-5 %)

tests/test_estimators_cpu.py::test_cpu_standin_is_pinned asserts the recorded values (and re-checks the outputs live
whenever the reference is importable).  No reference source text is stored."""
import json
import os
import platform
import sys
import time
import typing

sys.dont_write_bytecode = True
import typing_extensions  # noqa: E402

if not hasattr(typing, "Self"):
    typing.Self = typing_extensions.Self

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AFT_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.append(REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from src.config.schemas import ModelConfig as RefModelConfig, SystemConfig as RefSystemConfig  # noqa: E402  (reference)
from src.models import AdaFortiTranEstimator as RefAda, FortiTranEstimator as RefForti  # noqa: E402
import src as _ref_src  # noqa: E402

assert os.path.realpath(_ref_src.__file__).startswith(os.path.realpath(REF)), _ref_src.__file__

import adafortitran_amd as A  # noqa: E402
from adafortitran_amd import synth  # noqa: E402

SPEC = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4)
HID = (7, 42, 560)
B, ROUNDS, THREADS = 128, 9, int(os.environ.get("AFT_PIN_THREADS", "8"))


def build(cls, SC, MC, adaptive, sd):
    sc = SC(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    kw = dict(model_type="adafortitran" if adaptive else "fortitran", patch_size=(3, 2), num_layers=6, model_dim=128,
              num_head=4, max_seq_len=512, device="cpu")
    if adaptive:
        kw.update(channel_adaptivity_hidden_sizes=list(HID), adaptive_token_length=6)
    m = cls(sc, MC(**kw))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.eval()


def measure(adaptive):
    sd = synth.make_state_dict(**SPEC, adaptive_hidden=HID if adaptive else None, seed=20251114)
    ref = build(RefAda if adaptive else RefForti, RefSystemConfig, RefModelConfig, adaptive, sd)
    ours = build(A.AdaFortiTranEstimator if adaptive else A.FortiTranEstimator, A.SystemConfig, A.ModelConfig, adaptive, sd)
    inp = synth.make_inputs(B, seed=20251115)
    pil = torch.from_numpy(inp["pilots"])
    meta = synth.meta_tuple(inp) if adaptive else None
    call = (lambda m: m(pil, meta)) if adaptive else (lambda m: m(pil))
    t = {"reference": [], "standin": []}
    with torch.no_grad():
        a, b = call(ref), call(ours)
        diff = float((a - b).abs().max())
        for rnd in range(ROUNDS):
            order = [("reference", ref), ("standin", ours)]
            if rnd & 1:
                order.reverse()
            for name, m in order:
                t0 = time.perf_counter()
                call(m)
                t[name].append(time.perf_counter() - t0)
    med = {k: float(np.median(v)) for k, v in t.items()}
    return {"max_abs_diff": diff, "ymax": float(a.abs().max()), "median_s": med,
            "ratio_standin_over_reference": med["standin"] / med["reference"],
            "frames_per_s": {k: B / v for k, v in med.items()}}


if __name__ == "__main__":
    torch.set_num_threads(THREADS)
    rec = {"batch": B, "rounds": ROUNDS, "threads": THREADS, "torch": torch.__version__, "cpu": platform.processor() or "x86_64",
           "what": "imported reference (src.models.*Estimator) vs adafortitran_amd estimator on device='cpu', eval()+no_grad(), "
                   "same synthetic weights (synth seed 20251114) and inputs; interleaved rounds, alternating order, median",
           "adafortitran": measure(True), "fortitran": measure(False)}
    with open(os.path.join(HERE, "cpu_standin.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec, indent=1))
