#!/usr/bin/env python3
"""One encoder layer forward + backward against autograd (the body of tests/test_train_hip.py), printing the relative
error of every gradient.  AFT_TRAIN_ATTN_BWD_SPLIT=1 selects the two-pass attention backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adafortitran_amd import _abi  # noqa: E402
from adafortitran_amd import _lib   # switches change through the ABI (the library reads the environment once, at load)
from adafortitran_amd.training import HipEncoderLayerFunction, layer_params  # noqa: E402


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def dqkv_of(cfg, layer, x, gout, planes, d, two_pass):
    """dqkv [rows][3d] as the attention backward left it in the scratch buffer (offsets: plan_scratch, aft_train.hip)"""
    if two_pass:
        _lib.set_switch("AFT_TRAIN_ATTN_BWD_SPLIT", "1")
    else:
        _lib.set_switch("AFT_TRAIN_ATTN_BWD_SPLIT", None)
    grabbed = []
    real_empty = torch.empty

    def spy(*a, **k):
        t = real_empty(*a, **k)
        if k.get("dtype") == torch.uint8:
            grabbed.append(t)
        return t
    x.grad = None
    layer.zero_grad()
    out = HipEncoderLayerFunction.apply(x, cfg, float(os.environ.get("DROPOUT", "0")), 1234, *layer_params(layer))
    torch.empty = spy
    try:
        out.backward(gout)
    finally:
        torch.empty = real_empty
    torch.cuda.synchronize()
    rows = planes * cfg.tokens
    al = lambda n: (n + 63) // 64 * 64
    off = 3 * al(rows * d) + al(rows * 2 * d)
    sc = grabbed[-1].view(torch.float32)
    return sc[off:off + rows * 3 * d].view(planes, cfg.tokens, 3, d // 32, 32).clone()


def compare_dqkv(cfg, layer, x, gout, planes, d):
    a = dqkv_of(cfg, layer, x, gout, planes, d, False)
    a2 = dqkv_of(cfg, layer, x, gout, planes, d, False)
    b = dqkv_of(cfg, layer, x, gout, planes, d, True)
    b2 = dqkv_of(cfg, layer, x, gout, planes, d, True)
    print("one-pass run-to-run identical:", bool((a == a2).all()), " two-pass:", bool((b == b2).all()))
    tok_err = (a - b).abs().amax(dim=(0, 2, 3, 4))
    tok_ref = b.abs().amax(dim=(0, 2, 3, 4))
    worst = torch.argsort(tok_err / tok_ref.clamp_min(1e-30), descending=True)[:8]
    print("worst tokens (token, err, ref max):", [(int(t), float(tok_err[t]), float(tok_ref[t])) for t in worst])
    for blk, nm in enumerate("qkv"):
        da, db = a[:, :, blk], b[:, :, blk]
        err = (da - db).abs()
        print("dqkv", nm, "max err", float(err.max()), "ref max", float(db.abs().max()))
        bad = (err > 1e-4 * db.abs().max()).nonzero()
        if len(bad):
            print("   bad elements:", len(bad), "of", err.numel(), " first", bad[:4].tolist(), " last", bad[-2:].tolist())
            print("   bad planes", sorted(set(bad[:, 0].tolist())), "heads", sorted(set(bad[:, 2].tolist())))
            toks = sorted(set(bad[:, 1].tolist()))
            print("   bad tokens", toks[:12], "...", toks[-6:], "count", len(toks))
            print("   bad features", sorted(set(bad[:, 3].tolist())))


def main():
    d, heads, ofdm, planes, act = 128, 4, (120, 14), 6, "gelu"
    if len(sys.argv) > 1:
        d, heads, planes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
        ofdm = (int(sys.argv[4]), 14)
    cfg = _abi.make_config(ofdm=ofdm, pilot=(12, 2), patch=(3, 2), num_layers=1, model_dim=d, num_head=heads, activation=act)
    torch.manual_seed(0)
    layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=heads, dim_feedforward=2 * d, dropout=0.0, activation=act,
                                             batch_first=True).cuda().train()
    torch.manual_seed(1)
    x = torch.randn(planes, cfg.tokens, d, device="cuda", requires_grad=True)
    gout = torch.randn(planes, cfg.tokens, d, device="cuda")
    ref = layer(x)
    ref.backward(gout)
    ref_gx = x.grad.clone()
    ref_g = [p.grad.clone() for p in layer_params(layer)]
    x.grad = None
    layer.zero_grad()
    out = HipEncoderLayerFunction.apply(x, cfg, 0.0, 1234, *layer_params(layer))
    out.backward(gout)
    compare_dqkv(cfg, layer, x, gout, planes, d)
    print("tokens", cfg.tokens, "out", rel(out.detach(), ref.detach()), "gx", rel(x.grad, ref_gx))
    for name, p, rg in zip(_abi.LAYER_PARAM_NAMES, layer_params(layer), ref_g):
        print(f"  {name:24s} {rel(p.grad, rg):.3e}")
    w = layer_params(layer)[0].grad   # in_proj_weight [3d][d]
    rw = ref_g[0]
    for blk, nm in enumerate("qkv"):
        print("   in_proj", nm, rel(w[blk * d:(blk + 1) * d], rw[blk * d:(blk + 1) * d]))


if __name__ == "__main__":
    main()
