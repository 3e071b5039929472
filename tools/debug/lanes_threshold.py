#!/usr/bin/env python3
"""Where does running a forward as two lanes pay?  AFT_LANES=1 vs 2 over grids and batches, beside the forward's algorithmic GFLOP."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adafortitran_amd import _abi, synth
from adafortitran_amd import _lib   # switches change through the ABI (the library reads the environment once, at load)
from adafortitran_amd.hip_ops import engine_from_numpy
dev = lambda a: torch.from_numpy(a).cuda()
def gflop(spec, hid, B):
    S, T = spec["ofdm"]; p0, p1 = spec["patch"]; d, L = spec["model_dim"], spec["num_layers"]
    tokens = (S // p0) * (T // p1)
    row = L * (16 * d * d + 4 * tokens * d)
    return B * (2 * tokens * row + 2 * 2 * S * T * 9504) / 1e9
CASES = [(dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4), (7, 42, 560), (2, 4, 8, 12, 16, 24)),
         (dict(ofdm=(12, 14), pilot=(4, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4), (7, 42, 56), (32, 64, 128, 256, 512, 1024)),
         (dict(ofdm=(60, 14), pilot=(6, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4), (7, 42, 280), (8, 16, 32, 64, 128, 256)),
         (dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=64, num_head=2), (7, 42, 560), (8, 16, 32, 64, 128)),
         (dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=256, num_head=8), (7, 42, 560), (4, 8, 16, 32, 64)),
         (dict(ofdm=(240, 28), pilot=(24, 4), patch=(3, 2), num_layers=12, model_dim=256, num_head=8), (7, 42, 2240), (1, 2, 4, 8))]
for spec, hid, batches in CASES:
    sd = synth.make_state_dict(**spec, adaptive_hidden=hid, seed=1, max_seq_len=1120)
    cfg = _abi.make_config(**spec, adaptive_hidden=hid)
    eng = engine_from_numpy(cfg, sd, "cuda:0")
    for B in batches:
        inp = synth.make_inputs(B, ofdm=spec["ofdm"], pilot=spec["pilot"], seed=2)
        pil, meta = dev(inp["pilots"]), [dev(inp[k]) for k in ("snr", "ds", "dop")]
        res = {}
        for L in ("1", "2"):
            _lib.set_switch("AFT_LANES", L)
            for _ in range(10): eng.forward(pil, *meta)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                n = 50
                t0 = time.perf_counter()
                for _ in range(n): eng.forward(pil, *meta)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n)
            res[L] = best
        _lib.set_switch("AFT_LANES", None)
        tiles = (2 * B * eng.tokens + 31) // 32
        print(f"grid {spec['ofdm']} d={spec['model_dim']} B={B:5d}: {gflop(spec, hid, B):8.2f} GF, {tiles:6d} row tiles: one lane {res['1'] * 1e3:.4f} ms, two {res['2'] * 1e3:.4f} ms  -> {res['1'] / res['2']:.3f}x")
