"""Round 6: head dims 16 and 8 (8 / 16 heads at model_dim 128) -- the 16x16x4 attention kernel against round 5's 32x32x2 form (switches
AFT_ATTN_HD16_MFMA32 / AFT_ATTN_HD8_MFMA32): accuracy against the oracle and frames/s at 128 frames, interleaved."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from adafortitran_amd import _abi, _lib, synth
from adafortitran_amd.hip_ops import engine_from_numpy, profile_kernel
from oracle import oracle
oracle.build()
for d, heads in ((128, 8), (256, 16), (64, 4), (128, 16), (256, 32), (32, 4)):
    spec = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=d, num_head=heads)
    hid = (7, 42, 560)
    sd = synth.make_state_dict(**spec, adaptive_hidden=hid, seed=3, attn_gain=0.25, head_gain=2.0)
    cfg = _abi.make_config(**spec, adaptive_hidden=hid)
    eng = engine_from_numpy(cfg, sd, "cuda:0")
    B = 128
    inp = synth.make_inputs(B, seed=4)
    dev = lambda a: torch.from_numpy(a).cuda()
    args = (dev(inp["pilots"]), dev(inp["snr"]), dev(inp["ds"]), dev(inp["dop"]))
    ref = oracle.Oracle(cfg, sd).forward(inp["pilots"][:4], inp["snr"][:4], inp["ds"][:4], inp["dop"][:4])
    out = torch.empty((B, 120, 14), dtype=torch.complex64, device="cuda")
    res = {}
    for rnd in range(3):
        for sw in (None, "1"):
            _lib.set_switch("AFT_ATTN_HD16_MFMA32", sw); _lib.set_switch("AFT_ATTN_HD8_MFMA32", sw)
            for _ in range(20): eng.forward(*args, out=out)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): eng.forward(*args, out=out)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
            err = float(np.abs(out[:4].cpu().numpy() - ref).max() / np.abs(ref).max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            profile_kernel(eng, "attention", B, 2, None); e0.record(); profile_kernel(eng, "attention", B, 10, None); e1.record(); e1.synchronize()
            res.setdefault(sw, []).append((B / dt, err, e0.elapsed_time(e1) * 100))
    _lib.set_switch("AFT_ATTN_HD16_MFMA32", None); _lib.set_switch("AFT_ATTN_HD8_MFMA32", None)
    for sw, v in res.items():
        print(f"d={d} heads={heads} {'32x32x2' if sw else '16x16x4'}: {np.median([x[0] for x in v]):.0f} frames/s, attention {np.median([x[2] for x in v]):.1f} us, max|hip-oracle|/|y|max {max(x[1] for x in v):.2e}")
