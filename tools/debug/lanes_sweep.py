"""Round 6: per-frame rate of the default model (and config 5) at small batches for 1 .. 4 lanes (aft_set_switch AFT_LANES), interleaved rounds."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from adafortitran_amd import _abi, _lib, synth
from adafortitran_amd.hip_ops import engine_from_numpy
import bench

def rate(eng, pil, meta, out, steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): eng.forward(pil, *meta, out=out)
    torch.cuda.synchronize(); return pil.shape[0] * steps / (time.perf_counter() - t0)

for cname, batches in (("C3", (16, 32, 48, 64, 96, 128)), ("C5", (8, 16, 32))):
    c = bench.CONFIGS[cname]
    sd = synth.make_state_dict(**bench._spec(c), adaptive_hidden=c["hidden"], max_seq_len=c["max_seq_len"], seed=1)
    cfg = _abi.make_config(**bench._spec(c), adaptive_hidden=c["hidden"])
    eng = engine_from_numpy(cfg, sd, "cuda:0")
    base = None
    for B in batches:
        inp = synth.make_inputs(B, ofdm=c["ofdm"], pilot=c["pilot"], seed=2)
        dev = lambda a: torch.from_numpy(a).cuda()
        pil, meta = dev(inp["pilots"]), [dev(inp[k]) for k in ("snr", "ds", "dop")]
        out = torch.empty((B, *c["ofdm"]), dtype=torch.complex64, device="cuda")
        steps = 200 if cname == "C3" else 20
        res = {L: [] for L in (None, 1, 2, 3, 4)}
        for rnd in range(4):
            for L in res:
                _lib.set_switch("AFT_LANES", L)
                rate(eng, pil, meta, out, 20 if cname == "C3" else 3)
                res[L].append(rate(eng, pil, meta, out, steps))
        _lib.set_switch("AFT_LANES", None)
        med = {L: float(np.median(v)) for L, v in res.items()}
        print(cname, "B", B, " ".join(f"{'auto' if L is None else L}:{med[L]:.0f}" for L in med), "| best", max((L for L in med if L), key=lambda L: med[L]), flush=True)
