"""Training path of the encoder on the HIP device.

``HipEncoderLayerFunction`` is a ``torch.autograd.Function`` whose forward and backward are the
library's ``aft_encoder_layer_fwd_train_f32`` / ``aft_encoder_layer_bwd_f32``: one
``nn.TransformerEncoderLayer`` (post-LN, reference ``src/models/blocks/encoders.py:44-55``) in
``train()`` mode, with the activation tape owned by autograd.  ``encoder_stack_train`` chains it
over the layers of ``nn.TransformerEncoder`` so that ``loss.backward()`` in the reference's
``TrainingLoop.train_epoch`` (``src/main/trainer.py:195-233``) runs hand-written kernels for the
encoder (95 % of the FLOPs); ``HipConvEnhancerFunction``, ``HipLinearFunction`` and
``HipChannelAdapterFunction`` do the same for the conv stacks, the dense layers and the adapter MLPs.

Dropout uses a counter-based generator keyed by a per-call seed (drawn from torch's default
generator, so ``torch.manual_seed`` makes runs repeatable); the masks differ from PyTorch's Philox
stream, as any other implementation's would.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Sequence

import torch

from . import _abi, _lib


def flat_owner(p):
    from .optim import flat_owner as _fo   # optim imports nothing from here; imported lazily to keep module load light
    return _fo(p)


def direct_grad_ok(params) -> bool:
    """True when EVERY tensor in ``params`` is owned by a live ``optim.FlatParameters`` that asked for direct
    accumulation and its ``.grad`` still is that owner's float32 view.  Then the backward adds the parameter
    gradients straight into those views and reports "no gradient" to autograd, which saves autograd's own
    accumulate kernel per parameter.  The behaviour is scoped to the tagged parameters of that optimizer (the
    tag is an integer token that optim.py maps to its owner weakly: it dies with the FlatParameters): models driven by any other optimizer, with
    ``zero_grad(set_to_none=True)``, or whose ``.grad`` was re-pointed, get ordinary autograd gradients.
    ``FlatParameters(direct_accumulation=False)`` keeps autograd's AccumulateGrad path (needed for
    ``torch.autograd.grad``, ``backward(inputs=...)``, gradient hooks, torch DDP)."""
    for p in params:
        if p is None:
            continue
        owner = flat_owner(p)
        if owner is None or not owner.direct_accumulation or not owner.owns_grad(p):
            return False
    return True


def _layer_struct(cls, tensors: Sequence[torch.Tensor]):
    st = cls()
    for name, t in zip(_abi.LAYER_FIELDS, tensors):
        setattr(st, name, t.data_ptr())
    return st


class StackLink:
    """What a layer inside ``encoder_stack_train`` knows about its neighbours (passed in place of the bare seed): its own
    pre-allocated tape, whether the previous layer already left this layer's in-projection there, and the next layer's
    parameters + tape so that this layer's row-local kernel can compute the next in-projection as its tail
    (``aft_encoder_layer_fwd_train_chained_f32``).  Nothing here is differentiated: the in-projection's gradients still come
    from the layer that owns it, out of its own tape."""

    def __init__(self, seed, tape, qkv_ready=False, next_params=None, next_tape=None):
        self.seed, self.tape, self.qkv_ready = int(seed), tape, bool(qkv_ready)
        self.next_params, self.next_tape = next_params, next_tape
        self.next_qkv_written = False


class HipEncoderLayerFunction(torch.autograd.Function):
    """x [planes, tokens, d] -> layer(x); ``params`` are the layer's twelve tensors in
    ``_abi.LAYER_PARAM_NAMES`` order.  ``seed`` is an int, or a ``StackLink`` when the layer runs inside a stack."""

    @staticmethod
    def forward(ctx, x, cfg, dropout_p, seed, *params):
        lib = _lib.load()
        if x.dtype != torch.float32 or x.device.type != "cuda":
            raise ValueError("the HIP training path needs float32 tensors on the HIP device")
        x = x.contiguous()
        param_objs = params
        params = tuple(p.detach().contiguous() for p in params)
        planes = x.shape[0]
        batch = planes // 2
        link = seed if isinstance(seed, StackLink) else None
        if link is not None:
            seed, tape = link.seed, link.tape
        else:
            tape = torch.empty(lib.aft_encoder_tape_bytes(C.byref(cfg), batch), dtype=torch.uint8, device=x.device)
        scratch = torch.empty(lib.aft_encoder_train_scratch_bytes(C.byref(cfg), batch), dtype=torch.uint8, device=x.device)
        out = torch.empty_like(x)
        w = _layer_struct(_abi.AftLayerWeights, params)
        if link is None:
            _lib.check(lib.aft_encoder_layer_fwd_train_f32(
                C.byref(cfg), C.byref(w), x.data_ptr(), out.data_ptr(), tape.data_ptr(), tape.numel(),
                scratch.data_ptr(), scratch.numel(), batch, float(dropout_p), int(seed), _lib.current_stream_ptr(x.device)))
        else:
            nxt, wrote = None, C.c_int(0)
            if link.next_params is not None and link.next_tape is not None:
                next_params = tuple(p.detach().contiguous() for p in link.next_params)   # kept alive until the call returns
                nxt = _layer_struct(_abi.AftLayerWeights, next_params)
            _lib.check(lib.aft_encoder_layer_fwd_train_chained_f32(
                C.byref(cfg), C.byref(w), x.data_ptr(), out.data_ptr(), tape.data_ptr(), tape.numel(),
                scratch.data_ptr(), scratch.numel(), batch, float(dropout_p), int(seed), int(link.qkv_ready),
                C.byref(nxt) if nxt is not None else None, link.next_tape.data_ptr() if nxt is not None else None,
                link.next_tape.numel() if nxt is not None else 0, C.byref(wrote), _lib.current_stream_ptr(x.device)))
            link.next_qkv_written = bool(wrote.value)
        ctx.save_for_backward(x, tape, *params)
        ctx.cfg, ctx.dropout_p, ctx.seed, ctx.batch = cfg, float(dropout_p), int(seed), batch
        ctx.param_objs = param_objs
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x, tape, *params = ctx.saved_tensors
        cfg = ctx.cfg
        grad_out = grad_out.contiguous()
        direct = direct_grad_ok(ctx.param_objs)
        grads = [p.grad for p in ctx.param_objs] if direct else [torch.empty_like(p) for p in params]
        dx = torch.empty_like(x)
        scratch = torch.empty(lib.aft_encoder_train_scratch_bytes(C.byref(cfg), ctx.batch), dtype=torch.uint8, device=x.device)
        w = _layer_struct(_abi.AftLayerWeights, params)
        g = _layer_struct(_abi.AftLayerGrads, grads)
        _lib.check(lib.aft_encoder_layer_bwd_f32(
            C.byref(cfg), C.byref(w), x.data_ptr(), tape.data_ptr(), tape.numel(), grad_out.data_ptr(), dx.data_ptr(),
            C.byref(g), int(direct), scratch.data_ptr(), scratch.numel(), ctx.batch, ctx.dropout_p, ctx.seed,
            _lib.current_stream_ptr(x.device)))
        if direct:
            return (dx, None, None, None) + (None,) * len(params)
        return (dx, None, None, None, *grads)


def _ptr4(tensors):
    arr = (C.c_void_p * 4)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class HipConvEnhancerFunction(torch.autograd.Function):
    """ConvEnhancer (3x3 convs 1->8->32->8->1, ReLU after the first three; reference
    ``blocks/enhancers.py:5-31``) on x [N,1,S,T]: forward = the fused conv-stack kernel in its training
    variant (keeps the three hidden activations), backward = the same kernel run on the gradient with
    transposed, flipped weights (data gradient) + the weight/bias-gradient kernels."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, w4, b4):
        lib = _lib.load()
        x = x.contiguous()
        n, _, S, T = x.shape
        ws = [w.detach().contiguous() for w in (w1, w2, w3, w4)]
        bs = [b.detach().contiguous() for b in (b1, b2, b3, b4)]
        y = torch.empty_like(x)
        c1 = torch.empty((n, 8, T, S), dtype=torch.float32, device=x.device)
        c2 = torch.empty((n, 32, T, S), dtype=torch.float32, device=x.device)
        c3 = torch.empty((n, 8, T, S), dtype=torch.float32, device=x.device)
        scratch = torch.empty(lib.aft_conv_enhancer_fwd_scratch_bytes(n, S, T), dtype=torch.uint8, device=x.device)
        _lib.check(lib.aft_conv_enhancer_fwd_train_f32(C.byref(_ptr4(ws)), C.byref(_ptr4(bs)), x.data_ptr(), y.data_ptr(),
                                                       c1.data_ptr(), c2.data_ptr(), c3.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                                       n, S, T, _lib.current_stream_ptr(x.device)))
        ctx.save_for_backward(x, c1, c2, c3, *ws)
        ctx.param_objs = (w1, b1, w2, b2, w3, b3, w4, b4)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, c1, c2, c3, *ws = ctx.saved_tensors
        n, _, S, T = x.shape
        dy = dy.contiguous()
        objs = ctx.param_objs
        direct = direct_grad_ok(objs)
        grads = [p.grad for p in objs] if direct else [torch.empty_like(p) for p in objs]
        dx = torch.empty_like(x)
        scratch = torch.empty(lib.aft_conv_enhancer_scratch_bytes(n, S, T), dtype=torch.uint8, device=x.device)
        _lib.check(lib.aft_conv_enhancer_bwd_f32(C.byref(_ptr4(ws)), x.data_ptr(), c1.data_ptr(), c2.data_ptr(),
                                                 c3.data_ptr(), dy.data_ptr(), dx.data_ptr(), C.byref(_ptr4(grads[0::2])),
                                                 C.byref(_ptr4(grads[1::2])), int(direct), scratch.data_ptr(), scratch.numel(),
                                                 n, S, T, _lib.current_stream_ptr(x.device)))
        if direct:
            return (dx,) + (None,) * 8
        return (dx, *grads)


def _ptrs(tensors, n):
    arr = (C.c_void_p * n)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class HipChannelAdapterFunction(torch.autograd.Function):
    """ChannelAdapter (three MLPs 1 -> h0 -> h1 -> h2 on snr / delay spread / doppler, reference
    ``blocks/channel_adaptivity.py:24-63``) -> tokens [frames, h2/2, 6]; forward, data-gradient and parameter-
    gradient kernels of the library (``aft_adapter_fwd_train_f32`` / ``aft_adapter_bwd_f32``).  ``params`` are the
    18 tensors in the order {snr, ds, dop}_encoder.{0,2,4}.{weight, bias}."""

    @staticmethod
    def forward(ctx, snr, ds, dop, *params):
        lib = _lib.load()
        conds = [c.detach().reshape(-1).to(torch.float32).contiguous() for c in (snr, ds, dop)]
        ws = [p.detach().contiguous() for p in params[0::2]]
        bs = [p.detach().contiguous() for p in params[1::2]]
        frames, dev = conds[0].numel(), conds[0].device
        h0, h1, h2 = ws[0].shape[0], ws[1].shape[0], ws[2].shape[0]
        tokens = h2 // 2
        out = torch.empty((frames, tokens, 6), dtype=torch.float32, device=dev)
        a0 = torch.empty((frames, 3, h0), dtype=torch.float32, device=dev)
        a1 = torch.empty((frames, 3, h1), dtype=torch.float32, device=dev)
        hid = (C.c_int32 * 3)(h0, h1, h2)
        _lib.check(lib.aft_adapter_fwd_train_f32(C.byref(_ptrs(conds, 3)), C.byref(_ptrs(ws, 9)), C.byref(_ptrs(bs, 9)),
                                                 C.byref(hid), tokens, frames, out.data_ptr(), a0.data_ptr(), a1.data_ptr(),
                                                 _lib.current_stream_ptr(dev)))
        ctx.save_for_backward(a0, a1, *conds, *ws, *bs)
        ctx.param_objs = params
        return out

    @staticmethod
    def backward(ctx, dtok):
        lib = _lib.load()
        saved = ctx.saved_tensors
        a0, a1, conds, ws, bs = saved[0], saved[1], saved[2:5], saved[5:14], saved[14:23]
        frames, h0, h1, h2 = a0.shape[0], a0.shape[2], a1.shape[2], ws[2].shape[0]
        dtok = dtok.contiguous()
        objs = ctx.param_objs
        direct = direct_grad_ok(objs)
        grads = [p.grad for p in objs] if direct else [torch.empty_like(p) for p in objs]
        da0, da1 = torch.empty_like(a0), torch.empty_like(a1)
        hid = (C.c_int32 * 3)(h0, h1, h2)
        _lib.check(lib.aft_adapter_bwd_f32(C.byref(_ptrs(conds, 3)), C.byref(_ptrs(ws, 9)), C.byref(_ptrs(bs, 9)), C.byref(hid),
                                           h2 // 2, frames, a0.data_ptr(), a1.data_ptr(), dtok.data_ptr(), da0.data_ptr(),
                                           da1.data_ptr(), C.byref(_ptrs(grads[0::2], 9)), C.byref(_ptrs(grads[1::2], 9)),
                                           int(direct), _lib.current_stream_ptr(dtok.device)))
        if direct:
            return (None,) * (3 + len(objs))
        return (None, None, None, *grads)


class HipLinearFunction(torch.autograd.Function):
    """y = x W^T + b on [..., in] float32 tensors: the row-major MFMA GEMM forward, dgrad, split-K wgrad."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        w = weight.detach().contiguous()
        b = None if bias is None else bias.detach().contiguous()
        y = torch.empty((x2.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
        _lib.check(lib.aft_dense_fwd_f32(x2.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), y.data_ptr(),
                                         x2.shape[0], w.shape[1], w.shape[0], _lib.current_stream_ptr(x.device)))
        ctx.save_for_backward(x2, w)
        ctx.param_objs = (weight, bias)
        ctx.x_shape = x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, w = ctx.saved_tensors
        weight, bias = ctx.param_objs
        dy2 = dy.reshape(-1, w.shape[0]).contiguous()
        rows, out_f, in_f = x2.shape[0], w.shape[0], w.shape[1]
        direct = direct_grad_ok((weight, bias))
        dw = weight.grad if direct else torch.empty_like(w)
        db = None if bias is None else (bias.grad if direct else torch.empty_like(bias))
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        scratch = torch.empty(lib.aft_dense_bwd_scratch_bytes(rows, in_f, out_f), dtype=torch.uint8, device=dy.device)
        _lib.check(lib.aft_dense_bwd_f32(x2.data_ptr(), w.data_ptr(), dy2.data_ptr(), None if dx is None else dx.data_ptr(),
                                         dw.data_ptr(), None if db is None else db.data_ptr(), int(direct), scratch.data_ptr(),
                                         scratch.numel(), rows, in_f, out_f, _lib.current_stream_ptr(dy.device)))
        gx = None if dx is None else dx.view(ctx.x_shape)
        if direct:
            return gx, None, None
        return gx, dw, db


class HipLinear(torch.nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) whose grad-enabled float32 forward and backward on the
    HIP device run the library's GEMMs (HipLinearFunction).  The model's dense layers are thin (6..24 columns
    on one side), so they are memory-bound either way; measured 0.2-0.3 ms per training step faster than
    hipBLASLt's choices for these shapes.  ``lin.hip_training = False`` (or
    ``HipLinear.default_hip_training = False``) hands a layer back to PyTorch-ROCm."""

    default_hip_training = True
    hip_training = None   # None = follow the class default

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        on = HipLinear.default_hip_training if self.hip_training is None else self.hip_training
        if (on and x.device.type == "cuda" and torch.is_grad_enabled() and x.dtype == torch.float32
                and self.weight.dtype == torch.float32):
            return HipLinearFunction.apply(x, self.weight, self.bias)
        return super().forward(x)


class HipEmbedFunction(torch.autograd.Function):
    """conv_enhanced [P,S,T], adapter tokens [P,tokens,6] | None, linear_1.weight [d,K], linear_1.bias [d], positional table
    [1,max_len,d] (parameter or buffer) -> x0 [P,tokens,d] = linear_1(cat(PatchEmbedding(conv_enhanced), tokens)) + table[:, :tokens]
    as ONE launch, and its backward as one launch + the slice reductions (``aft_embed_fwd_train_f32`` / ``aft_embed_bwd_f32``;
    reference fortitran.py:212-217, blocks/encoders.py:67-68)."""

    @staticmethod
    def forward(ctx, conv, tok6, weight, bias, pos, patch):
        lib = _lib.load()
        conv = conv.contiguous()
        tok6 = None if tok6 is None else tok6.contiguous()
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        table = pos.detach()
        table = table[0] if table.dim() == 3 else table
        P, S, T = conv.shape
        p0, p1 = patch
        tokens, d = (S // p0) * (T // p1), w.shape[0]
        if not table.is_contiguous() or table.shape[0] < tokens or w.shape[1] != p0 * p1 + (0 if tok6 is None else 6):
            raise ValueError("embed: positional table shorter than the token count, or linear_1's width does not match the patch")
        x0 = torch.empty((P, tokens, d), dtype=torch.float32, device=conv.device)
        _lib.check(lib.aft_embed_fwd_train_f32(conv.data_ptr(), None if tok6 is None else tok6.data_ptr(), w.data_ptr(), b.data_ptr(),
                                               table.data_ptr(), x0.data_ptr(), P, S, T, p0, p1, d, _lib.current_stream_ptr(conv.device)))
        ctx.save_for_backward(conv, w, *(() if tok6 is None else (tok6,)))
        ctx.param_objs, ctx.dims = (weight, bias, pos), (P, S, T, p0, p1, d)
        return x0

    @staticmethod
    def backward(ctx, dx0):
        lib = _lib.load()
        conv, w, *rest = ctx.saved_tensors
        tok6 = rest[0] if rest else None
        weight, bias, pos = ctx.param_objs
        P, S, T, p0, p1, d = ctx.dims
        dx0 = dx0.contiguous()
        want_pos = bool(pos.requires_grad)
        direct = direct_grad_ok((weight, bias) + ((pos,) if want_pos else ()))
        dw = weight.grad if direct else torch.empty_like(w)
        db = bias.grad if direct else torch.empty_like(bias)
        dpos = None
        if want_pos:
            dpos = pos.grad if direct else torch.zeros_like(pos)   # the kernel writes the first `tokens` rows
        d_conv = torch.empty_like(conv)
        d_tok6 = None if tok6 is None else torch.empty_like(tok6)
        nbytes = lib.aft_embed_bwd_scratch_bytes(P, S, T, p0, p1, d, int(tok6 is not None))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=conv.device)
        _lib.check(lib.aft_embed_bwd_f32(conv.data_ptr(), None if tok6 is None else tok6.data_ptr(), w.data_ptr(), dx0.data_ptr(),
                                         d_conv.data_ptr(), None if d_tok6 is None else d_tok6.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                         None if dpos is None else dpos.data_ptr(), int(direct), scratch.data_ptr(), scratch.numel(),
                                         P, S, T, p0, p1, d, _lib.current_stream_ptr(conv.device)))
        if direct:
            return d_conv, d_tok6, None, None, None, None
        return d_conv, d_tok6, dw, db, dpos, None


class HipTailFunction(torch.autograd.Function):
    """x [P,tokens,d], linear_2.weight [p,d], linear_2.bias [p], conv_enhanced [P,S,T] -> conv_enhanced +
    InversePatchEmbedding(linear_2(x)) as ONE launch, backward one launch + the slice reductions (``aft_tail_fwd_train_f32`` /
    ``aft_tail_bwd_f32``; reference blocks/encoders.py:70, fortitran.py:225-227)."""

    @staticmethod
    def forward(ctx, x, weight, bias, resid, patch):
        lib = _lib.load()
        x, resid = x.contiguous(), resid.contiguous()
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        P, S, T = resid.shape
        p0, p1 = patch
        d = x.shape[-1]
        if w.shape != (p0 * p1, d) or x.shape[0] * x.shape[1] != P * (S // p0) * (T // p1):
            raise ValueError("tail: linear_2 / token count do not match the patch geometry")
        out = torch.empty_like(resid)
        _lib.check(lib.aft_tail_fwd_train_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), resid.data_ptr(), out.data_ptr(), P, S, T, p0, p1, d,
                                              _lib.current_stream_ptr(x.device)))
        ctx.save_for_backward(x, w)
        ctx.param_objs, ctx.dims = (weight, bias), (P, S, T, p0, p1, d)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        weight, bias = ctx.param_objs
        P, S, T, p0, p1, d = ctx.dims
        d_out = d_out.contiguous()
        direct = direct_grad_ok((weight, bias))
        dw = weight.grad if direct else torch.empty_like(w)
        db = bias.grad if direct else torch.empty_like(bias)
        dx = torch.empty_like(x)
        scratch = torch.empty(lib.aft_tail_bwd_scratch_bytes(P, S, T, p0, p1, d), dtype=torch.uint8, device=x.device)
        _lib.check(lib.aft_tail_bwd_f32(x.data_ptr(), w.data_ptr(), d_out.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), int(direct),
                                        scratch.data_ptr(), scratch.numel(), P, S, T, p0, p1, d, _lib.current_stream_ptr(x.device)))
        if direct:
            return dx, None, None, d_out, None
        return dx, dw, db, d_out, None


def layer_params(layer: torch.nn.Module):
    """The twelve parameter tensors of one nn.TransformerEncoderLayer, ABI order (_abi.LAYER_PARAM_NAMES).  Plain attribute reads:
    this runs for every layer of every step, and walking named_parameters() cost 0.2 ms of host time per step."""
    at = layer.self_attn
    return (at.in_proj_weight, at.in_proj_bias, at.out_proj.weight, at.out_proj.bias, layer.linear1.weight, layer.linear1.bias,
            layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias)


def encoder_stack_train(x: torch.Tensor, layers, cfg: _abi.AftConfig, dropout_p: float) -> torch.Tensor:
    """Run ``layers`` (an iterable of nn.TransformerEncoderLayer) in training mode on the HIP path."""
    layers = list(layers)
    seeds = torch.randint(0, 2 ** 62, (len(layers),), dtype=torch.int64).tolist()
    if _lib.get_switch("AFT_TRAIN_NO_QKV_CHAIN") or not layers:   # A/B switch: every layer runs its own in-projection GEMM
        for layer, seed in zip(layers, seeds):
            x = HipEncoderLayerFunction.apply(x, cfg, dropout_p, seed, *layer_params(layer))
        return x
    # Layers chained through their tapes: layer l's row-local kernel leaves layer l + 1's in-projection in layer l + 1's tape
    # (one GEMM launch per layer less; the product runs where the output tile is still in registers).
    lib = _lib.load()
    nbytes = lib.aft_encoder_tape_bytes(C.byref(cfg), x.shape[0] // 2)
    tapes = [torch.empty(nbytes, dtype=torch.uint8, device=x.device) for _ in layers]
    params = [layer_params(layer) for layer in layers]
    ready = False
    for i, seed in enumerate(seeds):
        last = i + 1 == len(layers)
        link = StackLink(seed, tapes[i], ready, None if last else params[i + 1], None if last else tapes[i + 1])
        x = HipEncoderLayerFunction.apply(x, cfg, dropout_p, link, *params[i])
        ready = link.next_qkv_written
    return x
