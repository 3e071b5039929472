#!/usr/bin/env python3
"""Plain NT product y = x W^T + b (aft_dense_fwd_f32) with 64-row and 96-row output tiles (switch AFT_GEMM_BM), interleaved:
us per launch and the fraction of the fp32 MFMA roof, over token-row counts and layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adafortitran_amd import _lib
lib = _lib.load()
dev = "cuda:0"
shapes = [(128, 384), (384, 128), (128, 128), (128, 256), (256, 128), (512, 512), (512, 1536), (512, 1024), (1024, 512), (256, 768)]
rows_list = [int(r) for r in os.environ.get("AFT_ROWS", "71680,35840,53760,17920,65520").split(",")]
for rows in rows_list:
    for in_f, out_f in shapes:
        x = torch.randn(rows, in_f, device=dev); w = torch.randn(out_f, in_f, device=dev); b = torch.randn(out_f, device=dev)
        y = torch.empty(rows, out_f, device=dev)
        st = _lib.current_stream_ptr(torch.device(dev))
        def run(n):
            for _ in range(n):
                _lib.check(lib.aft_dense_fwd_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, in_f, out_f, st))
        res = {}
        for rep in range(3):
            for bm in ("64", "96", None):
                _lib.set_switch("AFT_GEMM_BM", bm)
                run(3); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(20); e1.record(); e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 20
                res[bm] = min(res.get(bm, 1e9), us)
        gf = 2.0 * rows * in_f * out_f / 1e9
        print(f"rows={rows:6d} {in_f:4d}->{out_f:4d}  " + "  ".join(f"BM={k}: {v:7.1f} us ({gf / v / 157.3 * 1e3:.3f})" for k, v in res.items()), flush=True)
