#!/usr/bin/env python3
"""Compare the intermediate gradient buffers (g1 = d_o, g2, g2b, gff) of the fused and the unfused layer backward."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adafortitran_amd import _abi, _lib
from adafortitran_amd.training import layer_params, _layer_struct

lib = _lib.load()
d, heads, p, seed = 128, 4, float(sys.argv[1]) if len(sys.argv) > 1 else 0.1, 5
cfg = _abi.make_config(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=1, model_dim=d, num_head=heads)
torch.manual_seed(0)
layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=heads, dim_feedforward=2 * d, activation="gelu", dropout=p, batch_first=True).cuda().train()
params = [q.detach().contiguous() for q in layer_params(layer)]
planes = 2; batch = 1; rows = planes * cfg.tokens
x = torch.randn(planes, cfg.tokens, d, device="cuda"); gout = torch.randn_like(x); out = torch.empty_like(x)
tape = torch.zeros(lib.aft_encoder_tape_bytes(C.byref(cfg), batch), dtype=torch.uint8, device="cuda")
nscr = lib.aft_encoder_train_scratch_bytes(C.byref(cfg), batch)
w = _layer_struct(_abi.AftLayerWeights, params)
st = _lib.current_stream_ptr(x.device)
scr = torch.zeros(nscr, dtype=torch.uint8, device="cuda")
_lib.check(lib.aft_encoder_layer_fwd_train_f32(C.byref(cfg), C.byref(w), x.data_ptr(), out.data_ptr(), tape.data_ptr(), tape.numel(), scr.data_ptr(), nscr, batch, p, seed, st))
res = {}
for mode in ("fused", "unfused"):
    if mode == "unfused": os.environ["AFT_TRAIN_UNFUSED_BWD"] = "1"
    else: os.environ.pop("AFT_TRAIN_UNFUSED_BWD", None)
    scr = torch.zeros(nscr, dtype=torch.uint8, device="cuda")
    grads = [torch.zeros_like(q) for q in params]
    g = _layer_struct(_abi.AftLayerGrads, grads)
    dx = torch.empty_like(x)
    _lib.check(lib.aft_encoder_layer_bwd_f32(C.byref(cfg), C.byref(w), x.data_ptr(), tape.data_ptr(), tape.numel(), gout.data_ptr(), dx.data_ptr(), C.byref(g), 0, scr.data_ptr(), nscr, batch, p, seed, st))
    torch.cuda.synchronize()
    f = scr.view(torch.float32)
    n = (rows * d + 63) // 64 * 64
    res[mode] = {"g1(d_o)": f[0:rows * d].clone(), "g2": f[n:n + rows * d].clone(), "g2b": f[2 * n:2 * n + rows * d].clone(),
                 "gff": f[3 * n:3 * n + rows * 2 * d].clone(), "dx": dx.clone().view(-1)}
for k in res["fused"]:
    a, b = res["fused"][k], res["unfused"][k]
    nz_a, nz_b = float((a == 0).float().mean()), float((b == 0).float().mean())
    print(f"{k:8s} max|diff| {float((a - b).abs().max()):.3e}  |ref|max {float(b.abs().max()):.3e}  zero fraction fused {nz_a:.4f} unfused {nz_b:.4f}  mismatching zero pattern {float(((a == 0) != (b == 0)).float().mean()):.4f}")
