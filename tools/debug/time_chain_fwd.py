#!/usr/bin/env python3
"""Time one encoder layer's training forward (aft_encoder_layer_fwd_train_f32), fused vs unfused: [dropout]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adafortitran_amd import _abi, _lib
from adafortitran_amd.training import layer_params, _layer_struct
lib = _lib.load()
d, heads, p, seed = 128, 4, float(sys.argv[1]) if len(sys.argv) > 1 else 0.1, 5
cfg = _abi.make_config(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=1, model_dim=d, num_head=heads)
layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=heads, dim_feedforward=2 * d, activation="gelu", dropout=p, batch_first=True).cuda().train()
params = [q.detach().contiguous() for q in layer_params(layer)]
batch = 128; planes = 2 * batch
x = torch.randn(planes, cfg.tokens, d, device="cuda"); out = torch.empty_like(x)
tapes = [torch.zeros(lib.aft_encoder_tape_bytes(C.byref(cfg), batch), dtype=torch.uint8, device="cuda") for _ in range(6)]   # 6 tapes: no MALL reuse
nscr = lib.aft_encoder_train_scratch_bytes(C.byref(cfg), batch)
scr = torch.zeros(nscr, dtype=torch.uint8, device="cuda")
w = _layer_struct(_abi.AftLayerWeights, params); st = _lib.current_stream_ptr(x.device)
for mode in ("fused", "unfused"):
    if mode == "unfused": os.environ["AFT_TRAIN_UNFUSED_FWD"] = "1"
    else: os.environ.pop("AFT_TRAIN_UNFUSED_FWD", None)
    def run(i):
        t = tapes[i % 6]
        _lib.check(lib.aft_encoder_layer_fwd_train_f32(C.byref(cfg), C.byref(w), x.data_ptr(), out.data_ptr(), t.data_ptr(), t.numel(), scr.data_ptr(), nscr, batch, p, seed, st))
    for i in range(6): run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(24): run(i)
    e1.record(); e1.synchronize()
    print(f"dropout {p} layer forward {mode}: {e0.elapsed_time(e1) / 24 * 1e3:.1f} us")
