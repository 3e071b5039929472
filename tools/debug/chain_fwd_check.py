#!/usr/bin/env python3
"""Fused training forward chain vs the round-2 launch sequence: max relative difference of the layer output and of every tape tensor."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adafortitran_amd import _abi, _lib
from adafortitran_amd.training import layer_params, _layer_struct
lib = _lib.load()
d, heads = 128, 4
cfg = _abi.make_config(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=1, model_dim=d, num_head=heads)
for p, batch in ((0.0, 8), (0.1, 8), (0.0, 128)):
    torch.manual_seed(0)
    layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=heads, dim_feedforward=2 * d, activation="gelu", dropout=p, batch_first=True).cuda().train()
    params = [q.detach().contiguous() for q in layer_params(layer)]
    planes = 2 * batch; rows = planes * cfg.tokens
    x = torch.randn(planes, cfg.tokens, d, device="cuda")
    nt = lib.aft_encoder_tape_bytes(C.byref(cfg), batch); nscr = lib.aft_encoder_train_scratch_bytes(C.byref(cfg), batch)
    w = _layer_struct(_abi.AftLayerWeights, params); st = _lib.current_stream_ptr(x.device)
    res = {}
    for mode in ("fused", "unfused"):
        if mode == "unfused": _lib.set_switch("AFT_TRAIN_UNFUSED_FWD", "1")
        else: _lib.set_switch("AFT_TRAIN_UNFUSED_FWD", None)
        tape = torch.zeros(nt, dtype=torch.uint8, device="cuda"); scr = torch.zeros(nscr, dtype=torch.uint8, device="cuda"); out = torch.empty_like(x)
        _lib.check(lib.aft_encoder_layer_fwd_train_f32(C.byref(cfg), C.byref(w), x.data_ptr(), out.data_ptr(), tape.data_ptr(), nt, scr.data_ptr(), nscr, batch, p, 5, st))
        torch.cuda.synchronize()
        f = tape.view(torch.float32)
        al = lambda n: (n + 63) // 64 * 64
        off = 0; segs = {}
        for name, n in (("qkv", rows * 3 * d), ("attn", rows * d), ("lse", rows * heads), ("s1", rows * d), ("st1", rows * 2), ("x1", rows * d), ("a", rows * 2 * d), ("hd", rows * 2 * d), ("s2", rows * d), ("st2", rows * 2)):
            segs[name] = f[off:off + n].clone(); off += al(n)
        segs["out"] = out.view(-1).clone()
        res[mode] = segs
    print(f"p={p} B={batch}: " + "  ".join(f"{k} {float((res['fused'][k] - res['unfused'][k]).abs().max() / (res['unfused'][k].abs().max() + 1e-30)):.1e}" for k in res["fused"]))
