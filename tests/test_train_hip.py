"""Training path of the encoder: HIP forward/backward of one
nn.TransformerEncoderLayer against PyTorch autograd on the same device, fp32.

Tolerances: forward as the inference path (5e-5 of |y|max); gradients 2e-4 of each tensor's |g|max
(fp32 sums over 10^3..10^5 token rows in a different order than ATen's)."""
import numpy as np
import pytest
import torch

from adafortitran_amd import _abi

pytestmark = pytest.mark.gpu


def _cfg(d=128, heads=4, ofdm=(120, 14), act="gelu"):
    return _abi.make_config(ofdm=ofdm, pilot=(12, 2), patch=(3, 2), num_layers=1, model_dim=d, num_head=heads,
                            activation=act)


def _layer(d, heads, act, dropout, seed=0):
    torch.manual_seed(seed)
    layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=heads, dim_feedforward=2 * d, dropout=dropout,
                                             activation=act, batch_first=True).cuda()
    with torch.no_grad():   # non-trivial LayerNorm parameters and biases
        for n, p in layer.named_parameters():
            if "norm" in n or n.endswith("bias"):
                p.add_(0.1 * torch.randn_like(p))
    return layer


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("d,heads,ofdm,planes,act", [(128, 4, (120, 14), 6, "gelu"), (128, 4, (24, 14), 2, "relu"),
                                                     (256, 8, (48, 14), 4, "gelu"), (64, 2, (48, 14), 4, "gelu"), (192, 6, (48, 14), 2, "relu"),
                                                     # round 5: head dim 16 (zero-padded 32-feature heads), grids below 32
                                                     # tokens (one masked key tile)
                                                     (128, 8, (120, 14), 6, "gelu"), (256, 16, (48, 14), 2, "relu"), (64, 4, (24, 14), 4, "gelu"),
                                                     (128, 4, (12, 14), 6, "gelu"), (128, 8, (12, 14), 2, "relu"), (128, 4, (3, 14), 4, "gelu"),
                                                     (128, 4, (3, 2), 8, "gelu"), (192, 12, (24, 14), 2, "relu"),
                                                     # head dim 64: two 32-feature blocks per head (forward + two-pass backward)
                                                     (128, 2, (120, 14), 4, "gelu"), (64, 1, (48, 14), 4, "relu"), (256, 4, (24, 14), 2, "gelu"),
                                                     (128, 2, (12, 14), 6, "gelu"),
                                                     # every multiple of 32 (row-wise kernels with a half-filled last lane group)
                                                     (96, 3, (48, 14), 4, "gelu"), (32, 1, (48, 14), 2, "relu"), (160, 10, (24, 14), 2, "gelu"),
                                                     (224, 7, (24, 14), 2, "relu"), (32, 2, (12, 14), 4, "gelu"),
                                                     # late round 5: head dims 8 / 24 (padded to 32-feature heads), 40 / 48 (to 64)
                                                     (128, 16, (120, 14), 2, "gelu"), (96, 4, (48, 14), 4, "relu"), (192, 8, (24, 14), 2, "gelu"),
                                                     (160, 4, (48, 14), 2, "gelu"), (96, 2, (120, 14), 2, "relu"), (192, 4, (12, 14), 4, "gelu"),
                                                     # round 6, the general engine's shapes: model_dim up to 512 / off the multiples of 32, heads
                                                     # of 56 / 96 / 128 features (three / four 32-feature blocks) and of 25 / 20 / 15 (scalar re-lay)
                                                     (512, 8, (48, 14), 2, "gelu"), (512, 4, (24, 14), 2, "relu"), (384, 4, (48, 14), 2, "gelu"),
                                                     (256, 2, (120, 14), 2, "gelu"), (224, 4, (48, 14), 2, "relu"), (448, 8, (12, 14), 4, "gelu"),
                                                     (200, 8, (48, 14), 2, "gelu"), (80, 4, (24, 14), 4, "relu"), (120, 8, (48, 14), 2, "gelu"),
                                                     (288, 9, (24, 14), 2, "gelu"), (96, 1, (120, 14), 2, "relu"), (8, 1, (12, 14), 2, "gelu")])
def test_layer_forward_backward_matches_autograd(d, heads, ofdm, planes, act):
    from adafortitran_amd.training import HipEncoderLayerFunction, layer_params
    cfg = _cfg(d, heads, ofdm, act)
    layer = _layer(d, heads, act, 0.0).train()
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
    assert _rel(out.detach(), ref.detach()) <= 5e-5
    assert _rel(x.grad, ref_gx) <= 2e-4
    for name, p, rg in zip(_abi.LAYER_PARAM_NAMES, layer_params(layer), ref_g):
        assert _rel(p.grad, rg) <= 2e-4, name


@pytest.mark.parametrize("value", [float("nan"), 1e30])
def test_layer_training_bits_do_not_depend_on_stale_memory(value):
    """Tape, scratch and slice buffers come from torch.empty: the layer's output and every gradient must be the same bits
    whatever the allocator's pool held before (ragged 280-token planes, dropout on, d = 128 and 256)."""
    from adafortitran_amd.training import HipEncoderLayerFunction, layer_params

    def poison(v):
        blocks = [torch.full((n,), v, device="cuda") for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18) for _ in range(3)]
        del blocks

    for d, heads in ((128, 4), (256, 8)):
        cfg = _cfg(d, heads, (120, 14))
        layer = _layer(d, heads, "gelu", 0.1).train()
        torch.manual_seed(3)
        x0 = torch.randn(2, cfg.tokens, d, device="cuda")
        gout = torch.randn(2, cfg.tokens, d, device="cuda")

        def run():
            layer.zero_grad()
            x = x0.clone().requires_grad_(True)
            out = HipEncoderLayerFunction.apply(x, cfg, 0.1, 77, *layer_params(layer))
            out.backward(gout)
            return [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer_params(layer)]

        poison(0.0)
        ref = run()
        poison(value)
        got = run()
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.isfinite(a).all(), (d, i)
            assert torch.equal(a, b), (d, i)


@pytest.mark.parametrize("ofdm,heads,d", [((120, 14), 4, 128), ((24, 14), 4, 128), ((240, 28), 4, 128), ((48, 14), 8, 256)])
def test_one_pass_attention_backward_agrees_with_the_two_pass_form(ofdm, heads, d, switches):
    """attn_bwd_kernel (dQ, dK, dV in one pass, the default) against the two kernels it replaced
    (AFT_TRAIN_ATTN_BWD_SPLIT, read per launch): same masks, same products, sums in a different order --
    every gradient within 2e-6 of its tensor's |g|max; both forms bit-reproducible run to run.  Geometries: 9 key tiles
    (three full passes), 2 (one pass, one idle wave), 35 (twelve passes, the last with two waves; 13 KB of LDS tables),
    4 tiles of a 8-head layer; all ragged, dropout on."""
    from adafortitran_amd.training import HipEncoderLayerFunction, layer_params
    cfg = _cfg(d, heads, ofdm)
    layer = _layer(d, heads, "gelu", 0.1).train()
    torch.manual_seed(5)
    x0 = torch.randn(2, cfg.tokens, d, device="cuda")
    gout = torch.randn(2, cfg.tokens, d, device="cuda")

    def run():
        layer.zero_grad()
        x = x0.clone().requires_grad_(True)
        out = HipEncoderLayerFunction.apply(x, cfg, 0.1, 31, *layer_params(layer))
        out.backward(gout)
        return [x.grad.clone()] + [p.grad.clone() for p in layer_params(layer)]

    switches.unset("AFT_TRAIN_ATTN_BWD_SPLIT")
    one, one_again = run(), run()
    switches.set("AFT_TRAIN_ATTN_BWD_SPLIT", "1")
    two = run()
    for i, (a, b, c) in enumerate(zip(one, one_again, two)):
        assert torch.equal(a, b), i
        assert _rel(a, c) <= 2e-6, (i, _rel(a, c))
    assert any(not torch.equal(a, c) for a, c in zip(one, two))   # the switch did select another kernel
    # the launcher picks three- or twelve-wave workgroups (one or four problems each) by the problem count: same arithmetic per
    # problem, so the same bits (2 planes x heads problems here: a multiple of four)
    switches.unset("AFT_TRAIN_ATTN_BWD_SPLIT")
    switches.set("AFT_ATTN_BWD_GROUPS", "4")
    four = run()
    switches.set("AFT_ATTN_BWD_GROUPS", "1")
    three = run()
    switches.set("AFT_ATTN_BWD_GROUPS", "2")      # round 6: six-wave workgroups, two problems each (64 frames of the default model)
    six = run()
    for i, (a, b, c) in enumerate(zip(four, three, six)):
        assert torch.equal(a, b) and torch.equal(a, c), i


@pytest.mark.parametrize("d,heads,ofdm", [(128, 4, (120, 14)), (128, 4, (24, 14)), (256, 8, (48, 14))])
def test_stack_with_chained_in_projections_matches_the_unchained_stack(d, heads, ofdm, switches):
    """encoder_stack_train links the layers through their tapes: layer l's row-local kernel computes layer l + 1's
    in-projection (aft_encoder_layer_fwd_train_chained_f32).  Against the same stack with every layer running its own
    in-projection GEMM (AFT_TRAIN_NO_QKV_CHAIN): same seeds, same masks; output and every gradient within 2e-6 of the
    tensor's max (the product is summed in a different order), and against autograd as the layer test does.  d = 256 has
    no fused row-local kernel: the link must fall back to the GEMM (next_qkv_written = 0) and give identical bits."""
    from adafortitran_amd.training import encoder_stack_train, layer_params
    cfg = _cfg(d, heads, ofdm)
    layers = [_layer(d, heads, "gelu", 0.1, seed=10 + i).train() for i in range(3)]
    torch.manual_seed(8)
    x0 = torch.randn(4, cfg.tokens, d, device="cuda")
    gout = torch.randn(4, cfg.tokens, d, device="cuda")

    def run():
        for layer in layers:
            layer.zero_grad()
        x = x0.clone().requires_grad_(True)
        torch.manual_seed(99)                      # the per-layer dropout seeds are drawn from torch's generator
        out = encoder_stack_train(x, layers, cfg, 0.1)
        out.backward(gout)
        return [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for layer in layers for p in layer_params(layer)]

    switches.unset("AFT_TRAIN_NO_QKV_CHAIN")
    chained = run()
    switches.set("AFT_TRAIN_NO_QKV_CHAIN", "1")
    plain = run()
    for i, (a, b) in enumerate(zip(chained, plain)):
        if d == 256:
            assert torch.equal(a, b), i
        else:
            assert _rel(a, b) <= 2e-6, (i, _rel(a, b))
    if d != 256:
        assert any(not torch.equal(a, b) for a, b in zip(chained, plain))   # the link did change where the product runs

    # and against autograd through the same three nn.TransformerEncoderLayer modules, dropout off
    for layer in layers:
        layer.zero_grad()
    xr = x0.clone().requires_grad_(True)
    ref = xr
    for layer in layers:
        layer.dropout.p = layer.dropout1.p = layer.dropout2.p = layer.self_attn.dropout = 0.0
        ref = layer(ref)
    ref.backward(gout)
    ref_g = [xr.grad.clone()] + [p.grad.clone() for layer in layers for p in layer_params(layer)]
    switches.unset("AFT_TRAIN_NO_QKV_CHAIN")
    for layer in layers:
        layer.zero_grad()
    xh = x0.clone().requires_grad_(True)
    out = encoder_stack_train(xh, layers, cfg, 0.0)
    out.backward(gout)
    got = [xh.grad.clone()] + [p.grad.clone() for layer in layers for p in layer_params(layer)]
    assert _rel(out.detach(), ref.detach()) <= 5e-5
    for i, (a, b) in enumerate(zip(got, ref_g)):
        assert _rel(a, b) <= 3e-4, (i, _rel(a, b))


def test_dropout_is_consistent_between_forward_and_backward():
    """With p > 0 the layer is still a deterministic function of (x, seed): its backward must match a
    central finite difference of its forward along a random direction, and the keep rate must be 1-p."""
    from adafortitran_amd.training import HipEncoderLayerFunction, layer_params
    d, heads = 128, 4
    cfg = _cfg(d, heads, (24, 14))
    layer = _layer(d, heads, "gelu", 0.1).train()
    torch.manual_seed(2)
    x = torch.randn(2, cfg.tokens, d, device="cuda", dtype=torch.float32, requires_grad=True)
    gout = torch.randn_like(x)
    params = layer_params(layer)
    f = lambda xx: HipEncoderLayerFunction.apply(xx, cfg, 0.1, 99, *params)
    out = f(x)
    assert torch.equal(out, f(x))                       # same seed -> same masks
    assert not torch.equal(out, HipEncoderLayerFunction.apply(x, cfg, 0.1, 100, *params))
    out.backward(gout)
    v = torch.randn_like(x)
    eps = 1e-2
    with torch.no_grad():
        fd = ((f(x + eps * v).double() - f(x - eps * v).double()) * gout.double()).sum() / (2 * eps)
    an = (x.grad.double() * v.double()).sum()
    assert abs(float(fd - an)) <= 2e-3 * abs(float(an)) + 1e-3


def test_dropout_keep_rate():
    """Keep rate of the counter-based generator at the feed-forward site: all-zero input and unit
    linear1 bias make every hidden unit 1 before dropout; with unit linear2 weights the pre-norm sum
    of the second residual is (kept units)/(1-p) in every column, so its row mean / 2d is the kept
    fraction / (1-p)."""
    import ctypes as C
    from adafortitran_amd import _lib
    from adafortitran_amd.training import _layer_struct, layer_params
    d, heads, p = 128, 4, 0.25
    cfg = _cfg(d, heads, (24, 14), "relu")
    layer = _layer(d, heads, "relu", p)
    with torch.no_grad():
        for q in layer.parameters():
            q.zero_()
        layer.linear1.bias.fill_(1.0)
        layer.linear2.weight.fill_(1.0)
        layer.norm1.weight.fill_(1.0); layer.norm2.weight.fill_(1.0)
    lib = _lib.load()
    planes = 2
    x = torch.zeros(planes, cfg.tokens, d, device="cuda")
    out = torch.empty_like(x)
    tape = torch.zeros(lib.aft_encoder_tape_bytes(C.byref(cfg), 1), dtype=torch.uint8, device="cuda")
    scratch = torch.empty(lib.aft_encoder_train_scratch_bytes(C.byref(cfg), 1), dtype=torch.uint8, device="cuda")
    w = _layer_struct(_abi.AftLayerWeights, [q.detach() for q in layer_params(layer)])
    _lib.check(lib.aft_encoder_layer_fwd_train_f32(C.byref(cfg), C.byref(w), x.data_ptr(), out.data_ptr(), tape.data_ptr(),
                                                   tape.numel(), scratch.data_ptr(), scratch.numel(), 1, p, 7, None))
    torch.cuda.synchronize()
    # the hidden activation after dropout is the 8th block of the tape: qkv, attn, lse, s1, st1, x1, a, hd
    rows, ff = planes * cfg.tokens, 2 * d
    al = lambda n: (n + 63) // 64 * 64
    hd_off = al(rows * 3 * d) + al(rows * d) + al(rows * heads) + al(rows * d) + al(rows * 2) + al(rows * d) + al(rows * ff)
    hd = tape.view(torch.float32)[hd_off:hd_off + rows * ff]
    kept = float((hd != 0).double().mean())
    assert abs(kept - (1 - p)) < 0.01
    vals = hd[hd != 0]
    assert torch.allclose(vals, torch.full_like(vals, 1 / (1 - p)))


def _model(name, dropout, pos="learnable"):
    import adafortitran_amd as A
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    kw = dict(model_type=name, patch_size=(3, 2), num_layers=2, model_dim=128, num_head=4, activation="gelu",
              max_seq_len=512, pos_encoding_type=pos, device="cuda", dropout=dropout)
    if name == "adafortitran":
        kw.update(channel_adaptivity_hidden_sizes=[7, 42, 560], adaptive_token_length=6)
    cls = A.AdaFortiTranEstimator if name == "adafortitran" else A.FortiTranEstimator
    return cls(sc, A.ModelConfig(**kw))


@pytest.mark.parametrize("name,pos", [("fortitran", "learnable"), ("adafortitran", "learnable"), ("adafortitran", "sinusoidal")])
def test_full_model_training_step_matches_pytorch_autograd(name, pos):
    """loss.backward() through the whole estimator: encoder on the HIP training kernels vs the same
    module differentiated entirely by PyTorch-ROCm, dropout 0, identical parameters and batch.  (sinusoidal: the positional table is
    a buffer -- the fused embedding end is asked for no table gradient.)"""
    from adafortitran_amd import synth
    torch.manual_seed(0)
    model = _model(name, 0.0, pos).train()
    B = 6
    inp = synth.make_inputs(B, seed=5)
    pil = torch.from_numpy(inp["pilots"]).cuda()
    tgt = torch.from_numpy(inp["target"]).cuda()
    meta = synth.meta_tuple(inp) if name == "adafortitran" else None
    call = (lambda: model(pil, meta)) if meta is not None else (lambda: model(pil))

    def step(hip):
        from adafortitran_amd import training
        model.transformer_encoder.hip_training = hip
        model.initial_enhancer.hip_training = model.final_refiner.hip_training = hip
        training.HipLinear.default_hip_training = hip
        if hasattr(model, "channel_adapter"):
            model.channel_adapter.hip_training = hip
        model.zero_grad()
        out = call()
        loss = torch.nn.functional.mse_loss(torch.view_as_real(out), torch.view_as_real(tgt))
        loss.backward()
        return float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    loss_ref, g_ref = step(False)
    loss_hip, g_hip = step(True)
    assert abs(loss_hip - loss_ref) <= 1e-5 * abs(loss_ref)
    assert g_ref.keys() == g_hip.keys()
    for n in g_ref:
        # (the adapter feeds raw conditions -- Doppler 1400, delay spread 350 -- through three MLPs: every fp32 evaluation of its
        #  gradients carries ~1e-4 .. 1e-3 of chaotic noise, tests/test_train_golden.py)
        assert _rel(g_hip[n], g_ref[n]) <= (2e-3 if n.startswith("channel_adapter") else 5e-4), n


def test_training_reduces_the_loss_with_dropout():
    """Adam steps with dropout 0.1 on the HIP encoder path (reference train_epoch, trainer.py:195-233)
    towards the output of a fixed teacher network (itself run by the HIP inference path)."""
    from adafortitran_amd import synth
    torch.manual_seed(1)
    teacher = _model("fortitran", 0.0).eval()
    torch.manual_seed(0)
    model = _model("fortitran", 0.1).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    pil = torch.from_numpy(synth.make_inputs(8, seed=3)["pilots"]).cuda()
    with torch.no_grad():
        tgt = teacher(pil)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(torch.view_as_real(model(pil)), torch.view_as_real(tgt))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < 0.5 * losses[0], losses


def test_fused_adam_matches_torch_adam_on_device():
    from adafortitran_amd.optim import ShardedFlatAdam
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.GELU(), torch.nn.Linear(300, 7)).cuda()
    b = torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.GELU(), torch.nn.Linear(300, 7)).cuda()
    b.load_state_dict(a.state_dict())
    x, y = torch.randn(32, 64, device="cuda"), torch.randn(32, 7, device="cuda")
    ref = torch.optim.Adam(a.parameters(), lr=3e-3, weight_decay=1e-3)
    opt = ShardedFlatAdam(b.parameters(), lr=3e-3, weight_decay=1e-3)
    for _ in range(6):
        for net, o in ((a, ref), (b, opt)):
            o.zero_grad()
            torch.nn.functional.mse_loss(net(x), y).backward()
            o.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=1e-6)


def test_direct_gradient_accumulation_matches_autograd_accumulation():
    """ShardedFlatAdam lets the encoder backward add into the flat .grad views; same gradients as the
    autograd-accumulated path."""
    from adafortitran_amd import synth, training
    from adafortitran_amd.optim import FlatParameters
    torch.manual_seed(0)
    model = _model("fortitran", 0.0).train()
    pil = torch.from_numpy(synth.make_inputs(4, seed=9)["pilots"]).cuda()
    tgt = torch.from_numpy(synth.make_inputs(4, seed=9)["target"]).cuda()

    def grads(direct):
        flat.direct_accumulation = direct
        flat.zero_grad()
        torch.nn.functional.mse_loss(torch.view_as_real(model(pil)), torch.view_as_real(tgt)).backward()
        return flat.grad.clone()

    flat = FlatParameters(model.parameters())
    assert not training.direct_grad_ok(model.parameters())             # off unless the owner asks for it
    a, b = grads(False), grads(True)
    assert training.direct_grad_ok(model.parameters())
    assert _rel(b, a) <= 1e-6
    # scoped to the tagged parameters of a LIVE owner: another model, a re-pointed .grad or a released owner
    # all get ordinary autograd gradients
    other = _model("fortitran", 0.0).train()
    assert not training.direct_grad_ok(other.parameters())
    w = model.pilot_upsampler.weight
    keep = w.grad
    w.grad = torch.zeros_like(w)
    assert not training.direct_grad_ok([w])
    w.grad = keep
    flat.release()
    assert not training.direct_grad_ok(model.parameters())
    model.zero_grad(set_to_none=True)
    torch.nn.functional.mse_loss(torch.view_as_real(model(pil)), torch.view_as_real(tgt)).backward()
    got = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    want = torch.cat([a[o:o + p.numel()] for p, o in zip(flat.params, flat.offsets)])
    assert _rel(got, want) <= 1e-6


def test_eval_after_training_uses_the_updated_flat_parameters():
    """ShardedFlatAdam re-homes the parameters into one flat buffer; the inference engine must follow
    (new pointers, in-place updates): HIP eval output == PyTorch composite on the same module."""
    from adafortitran_amd import synth
    from adafortitran_amd.optim import ShardedFlatAdam
    torch.manual_seed(3)
    model = _model("adafortitran", 0.1)
    inp = synth.make_inputs(6, seed=11)
    pil, tgt = torch.from_numpy(inp["pilots"]).cuda(), torch.from_numpy(inp["target"]).cuda()
    meta = synth.meta_tuple(inp)
    model.eval()
    with torch.no_grad():
        before = model(pil, meta).clone()          # builds the engine on the original parameter tensors
    opt = ShardedFlatAdam(model.parameters(), lr=1e-3)
    model.train()
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(torch.view_as_real(model(pil, meta)), torch.view_as_real(tgt)).backward()
        opt.step()
    model.eval()
    with torch.no_grad():
        hip = model(pil, meta)                     # HIP inference path
    model.transformer_encoder.hip_training = False
    try:
        ref = model(pil, meta).detach()            # grad-enabled eval forward: pure PyTorch composite
    finally:
        model.transformer_encoder.hip_training = True
    assert not torch.allclose(torch.view_as_real(hip), torch.view_as_real(before))
    assert _rel(torch.view_as_real(hip), torch.view_as_real(ref)) <= 5e-5


@pytest.mark.parametrize("S,T,n", [(120, 14, 6), (24, 4, 3), (150, 8, 2), (240, 28, 2)])
def test_conv_enhancer_forward_backward_matches_autograd(S, T, n):
    """The conv stack's training forward/backward (fused MFMA kernel, its transposed run for the data
    gradient, MFMA weight gradients) against PyTorch-ROCm autograd (MIOpen) on the same module."""
    import adafortitran_amd.blocks as blocks
    torch.manual_seed(S + T)
    enh = blocks.ConvEnhancer().cuda()
    x = torch.randn(n, 1, S, T, device="cuda", requires_grad=True)
    gy = torch.randn(n, 1, S, T, device="cuda")
    params = list(enh.parameters())

    def run(hip):
        enh.hip_training = hip
        enh.zero_grad(); x.grad = None
        y = enh(x)
        y.backward(gy)
        return y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in params]

    y0, gx0, g0 = run(False)
    y1, gx1, g1 = run(True)
    assert _rel(y1, y0) <= 2e-5
    assert _rel(gx1, gx0) <= 1e-4
    for (name, _), a, b in zip(enh.named_parameters(), g1, g0):
        assert _rel(a, b) <= 2e-4, name


@pytest.mark.parametrize("n", [3, 10, 130])
def test_training_conv_column_ranges_reproduce_whole_planes(n, switches):
    """With fewer planes than CUs the training conv kernel (conv_stream_kernel<0, true>, forward and data gradient) splits every
    plane of the default grid into 2 or 4 column ranges that recompute their neighbours' edge columns (64 frames -- the reference's
    default batch -- are 128 planes on 256 CUs): the output, the three saved activations' consumers (the data gradient and every
    weight gradient) carry the same BITS whatever the split."""
    import adafortitran_amd.blocks as blocks
    torch.manual_seed(n)
    enh = blocks.ConvEnhancer().cuda()
    x = torch.randn(n, 1, 120, 14, device="cuda", requires_grad=True)
    gy = torch.randn(n, 1, 120, 14, device="cuda")
    params = list(enh.parameters())

    def run(split):
        if split is None:
            switches.unset("AFT_CONV_NSPLIT")
        else:
            switches.set("AFT_CONV_NSPLIT", split)
        enh.zero_grad(); x.grad = None
        y = enh(x)
        y.backward(gy)
        return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in params]

    ref = run("1")
    for split in ("2", "4", None):
        got = run(split)
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a, b), (split, i)


@pytest.mark.parametrize("rows,in_f,out_f,bias", [(560, 12, 128, True), (561, 6, 128, True), (1120, 128, 6, True),
                                                   (37, 1, 7, True), (128, 42, 560, False), (256, 24, 1680, True),
                                                   # the aligned fast path (gemm_fast_kernel) with ragged row / column tiles
                                                   (1008, 64, 144, False), (1000, 128, 132, True), (2064, 48, 200, True)])
def test_dense_layer_forward_backward_matches_autograd(rows, in_f, out_f, bias):
    """HipLinear on the layer shapes of the model (including row lengths that are not a multiple of 4) and on shapes
    that take the aligned GEMM path with partial 64-row / 128-column tiles."""
    from adafortitran_amd.training import HipLinear
    torch.manual_seed(rows + in_f)
    lin = HipLinear(in_f, out_f, bias=bias).cuda()
    x = torch.randn(rows, in_f, device="cuda", requires_grad=True)
    gy = torch.randn(rows, out_f, device="cuda")

    def run(hip):
        lin.hip_training = hip
        lin.zero_grad(); x.grad = None
        y = lin(x)
        y.backward(gy)
        return y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in lin.parameters()]

    y0, gx0, g0 = run(False)
    y1, gx1, g1 = run(True)
    assert _rel(y1, y0) <= 1e-5 and _rel(gx1, gx0) <= 1e-5
    for a, b in zip(g1, g0):
        assert _rel(a, b) <= 1e-4


@pytest.mark.parametrize("rows,in_f,out_f", [(71680, 128, 384), (71680, 384, 128), (1000, 128, 132), (2064, 48, 200), (97, 64, 128),
                                              (35840, 128, 256)])
def test_dense_layer_96_row_tiles_have_the_64_row_tiles_bits(rows, in_f, out_f, switches):
    """Round 6: the plain NT / NN products pick 96-row output tiles (three workgroups per CU) where that saves a round of tiles over the
    resident workgroups (71 680 token rows: 747 tiles on 768 workgroups instead of 1 120 on 1 024).  Every output element sums its k
    steps in the same order in both tilings: forced either way (switch AFT_GEMM_BM) or chosen, the SAME bits -- full, ragged row and
    ragged column tiles."""
    from adafortitran_amd.training import HipLinear
    torch.manual_seed(rows + in_f)
    lin = HipLinear(in_f, out_f, bias=True).cuda()
    lin.hip_training = True
    x = torch.randn(rows, in_f, device="cuda", requires_grad=True)
    gy = torch.randn(rows, out_f, device="cuda")

    def run(bm):
        if bm is None:
            switches.unset("AFT_GEMM_BM")
        else:
            switches.set("AFT_GEMM_BM", bm)
        lin.zero_grad(); x.grad = None
        y = lin(x)
        y.backward(gy)
        return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in lin.parameters()]

    ref = run("64")
    for bm in ("96", None):
        got = run(bm)
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a, b), (bm, i)
    lin.hip_training = False
    lin.zero_grad(); x.grad = None
    y0 = lin(x); y0.backward(gy)
    assert _rel(ref[0], y0.detach()) <= 1e-5 and _rel(ref[1], x.grad) <= 1e-5


ENDS_CASES = [((120, 14), (3, 2), 128, True, 4), ((120, 14), (3, 2), 128, False, 2), ((24, 14), (3, 2), 512, True, 2),
              ((48, 16), (4, 4), 64, True, 6), ((96, 14), (12, 2), 200, False, 2), ((12, 14), (3, 2), 128, True, 256),
              ((20, 6), (1, 1), 8, True, 2), ((64, 8), (8, 4), 256, True, 4)]


@pytest.mark.parametrize("ofdm,patch,d,adaptive,planes", ENDS_CASES)
def test_fused_embedding_end_matches_autograd(ofdm, patch, d, adaptive, planes):
    """training.HipEmbedFunction (patch embedding + adapter concat + linear_1 + positional table: one launch each way) against the
    PyTorch composite it replaces (reference fortitran.py:212-217, encoders.py:67-68): output, d(conv_enhanced), d(adapter tokens),
    dW1, db1 and the positional table's gradient (rows beyond the token count stay zero).  Grids with partial token blocks, patches of
    1 / 6 / 16 / 24 / 32 elements, model dims 8 .. 512, 2 .. 256 planes."""
    from adafortitran_amd.blocks import PatchEmbedding
    from adafortitran_amd.training import HipEmbedFunction
    torch.manual_seed(ofdm[0] + d)
    S, T = ofdm
    p = patch[0] * patch[1]
    tokens, K = (S // patch[0]) * (T // patch[1]), p + (6 if adaptive else 0)
    conv = torch.randn(planes, S, T, device="cuda", requires_grad=True)
    tok6 = torch.randn(planes, tokens, 6, device="cuda", requires_grad=True) if adaptive else None
    w = (torch.randn(d, K, device="cuda") / K ** 0.5).requires_grad_(True)
    b = torch.randn(d, device="cuda", requires_grad=True)
    pos = torch.randn(1, tokens + 5, d, device="cuda", requires_grad=True)
    gy = torch.randn(planes, tokens, d, device="cuda")
    leaves = [t for t in (conv, tok6, w, b, pos) if t is not None]

    def grads(y):
        for t in leaves:
            t.grad = None
        y.backward(gy)
        return [t.grad.clone() for t in leaves]

    tk = PatchEmbedding(patch)(conv)
    if adaptive:
        tk = torch.cat((tk, tok6), dim=2)
    y0 = torch.nn.functional.linear(tk, w, b) + pos[:, :tokens]
    g0 = grads(y0)
    y1 = HipEmbedFunction.apply(conv, tok6, w, b, pos, patch)
    g1 = grads(y1)
    assert _rel(y1.detach(), y0.detach()) <= 2e-6
    for a, c in zip(g1, g0):
        assert a.shape == c.shape and _rel(a, c) <= 2e-5, _rel(a, c)
    assert torch.count_nonzero(g1[-1][:, tokens:]) == 0
    # a table that carries no gradient (the sinusoid's buffer): same output, no table gradient asked for
    y2 = HipEmbedFunction.apply(conv, tok6, w, b, pos.detach(), patch)
    assert torch.equal(y2.detach(), y1.detach())
    for t in leaves:
        t.grad = None
    y2.backward(gy)
    assert pos.grad is None and torch.equal(conv.grad, g1[0]) and torch.equal(w.grad, g1[-3])


@pytest.mark.parametrize("ofdm,patch,adaptive,planes", [((120, 14), (3, 2), True, 4), ((120, 14), (3, 2), False, 2), ((12, 14), (3, 2), True, 256),
                                                         ((120, 14), (3, 2), True, 256), ((48, 16), (4, 4), False, 6), ((60, 14), (5, 2), True, 18)])
def test_embedding_backward_on_mfmas_has_the_vector_kernels_bits(ofdm, patch, adaptive, planes, switches):
    """The default model's embedding backward (model_dim 128, at most 16 input features) runs both of its products on 16x16x4 MFMAs
    (embed_rows_bwd128_kernel); switch AFT_EMBED_BWD_GENERIC keeps the vector-ALU kernel every other shape runs.  An fp32 MFMA is an
    FMA chain in k order and the two kernels visit c / the rows in the same order: every output -- d(conv_enhanced),
    d(adapter tokens), dW1, db1, the table's gradient -- has the same bits.  Full and partial token blocks, 2 .. 256 planes (one to many
    passes per workgroup, a last pass with fewer than four planes), 6 / 10 / 12 / 16 input features."""
    from adafortitran_amd.training import HipEmbedFunction
    torch.manual_seed(planes + ofdm[0])
    S, T = ofdm
    p, d = patch[0] * patch[1], 128
    tokens, K = (S // patch[0]) * (T // patch[1]), p + (6 if adaptive else 0)
    conv = torch.randn(planes, S, T, device="cuda", requires_grad=True)
    tok6 = torch.randn(planes, tokens, 6, device="cuda", requires_grad=True) if adaptive else None
    w = (torch.randn(d, K, device="cuda") / K ** 0.5).requires_grad_(True)
    b = torch.randn(d, device="cuda", requires_grad=True)
    pos = torch.randn(1, tokens + 3, d, device="cuda", requires_grad=True)
    gy = torch.randn(planes, tokens, d, device="cuda")
    leaves = [t for t in (conv, tok6, w, b, pos) if t is not None]

    def grads():
        for t in leaves:
            t.grad = None
        HipEmbedFunction.apply(conv, tok6, w, b, pos, patch).backward(gy)
        return [t.grad.clone() for t in leaves]

    switches.unset("AFT_EMBED_BWD_GENERIC")
    fast = grads()
    switches.set("AFT_EMBED_BWD_GENERIC", "1")
    generic = grads()
    for i, (x, y) in enumerate(zip(fast, generic)):
        assert torch.isfinite(x).all() and torch.equal(x, y), i


@pytest.mark.parametrize("ofdm,patch,d,adaptive,planes", ENDS_CASES)
def test_fused_reconstruction_end_matches_autograd(ofdm, patch, d, adaptive, planes):
    """training.HipTailFunction (linear_2 + inverse patch embedding + the residual: one launch each way) against the PyTorch composite
    (reference encoders.py:70, fortitran.py:225-227): output, dx, dW2, db2 and the residual's gradient."""
    from adafortitran_amd.blocks import InversePatchEmbedding
    from adafortitran_amd.training import HipTailFunction
    torch.manual_seed(ofdm[1] + d)
    S, T = ofdm
    p = patch[0] * patch[1]
    tokens = (S // patch[0]) * (T // patch[1])
    x = torch.randn(planes, tokens, d, device="cuda", requires_grad=True)
    w = (torch.randn(p, d, device="cuda") / d ** 0.5).requires_grad_(True)
    b = torch.randn(p, device="cuda", requires_grad=True)
    resid = torch.randn(planes, S, T, device="cuda", requires_grad=True)
    gy = torch.randn(planes, S, T, device="cuda")
    leaves = [x, w, b, resid]

    def grads(y):
        for t in leaves:
            t.grad = None
        y.backward(gy)
        return [t.grad.clone() for t in leaves]

    y0 = resid + InversePatchEmbedding(ofdm, patch)(torch.nn.functional.linear(x, w, b))
    g0 = grads(y0)
    y1 = HipTailFunction.apply(x, w, b, resid, patch)
    g1 = grads(y1)
    assert _rel(y1.detach(), y0.detach()) <= 2e-6
    for a, c in zip(g1, g0):
        assert a.shape == c.shape and _rel(a, c) <= 2e-5, _rel(a, c)


def test_fused_ends_are_what_the_training_step_runs(switches):
    """The estimator's grad-enabled forward hands the encoder the conv-enhanced planes (fused ends) by default; with the switch
    AFT_TRAIN_NO_FUSED_ENDS it runs PyTorch's unfold / cat / add around HipLinear as before.  The fused kernels keep the rounding
    sequence of the launches they replace, so the loss and EVERY gradient that flows through them -- conv stacks, adapter, encoder
    layers, pilot_upsampler -- have the same BITS either way; only the five parameter gradients the fused kernels sum themselves
    (linear_1 weight / bias, the positional table, linear_2 weight / bias) come out in another summation order."""
    import adafortitran_amd as A
    from adafortitran_amd import synth
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    torch.manual_seed(3)
    model = A.AdaFortiTranEstimator(sc, A.ModelConfig(model_type="adafortitran", patch_size=(3, 2), num_layers=2, model_dim=128, num_head=4,
                                                      device="cuda", dropout=0.1, channel_adaptivity_hidden_sizes=[7, 42, 560],
                                                      adaptive_token_length=6)).train()
    inp = synth.make_inputs(4, seed=10)
    pil, tgt = torch.from_numpy(inp["pilots"]), torch.from_numpy(inp["target"]).to("cuda")
    meta = synth.meta_tuple(inp)

    def run():
        model.zero_grad()
        torch.manual_seed(77)            # the same dropout seeds in both runs
        loss = torch.view_as_real(model(pil, meta) - tgt).pow(2).mean()
        loss.backward()
        return {"loss": loss.detach().clone(), **{n: p.grad.clone() for n, p in model.named_parameters()}}

    probe = torch.zeros(8, 120, 14, device="cuda")
    enc = model.transformer_encoder
    with torch.enable_grad():
        assert enc.fused_ends_ok(probe, (3, 2))
        fused = run()
        switches.set("AFT_TRAIN_NO_FUSED_ENDS", "1")
        assert not enc.fused_ends_ok(probe, (3, 2))
        apart = run()
    with torch.no_grad():
        switches.unset("AFT_TRAIN_NO_FUSED_ENDS")
        assert not enc.fused_ends_ok(probe, (3, 2))          # inference never takes the training kernels
    own = {"transformer_encoder.linear_1.weight", "transformer_encoder.linear_1.bias", "transformer_encoder.linear_2.weight",
           "transformer_encoder.linear_2.bias", "transformer_encoder.positional_encoding.position_embeddings"}
    for n in fused:
        if n in own:
            assert _rel(fused[n], apart[n]) <= 2e-5, (n, _rel(fused[n], apart[n]))
        else:
            assert torch.equal(fused[n], apart[n]), n


@pytest.mark.parametrize("batch,reps", [(128, 40), (64, 40), (6, 60)])
def test_training_step_soak_same_bits_every_time(batch, reps):
    """The whole training step (dropout on, same seeds) run `reps` times back to back on one stream: loss and every gradient must come
    out with the SAME BITS every time -- the hand-overs inside the kernels (LDS exchanges, wave-private regions, the slice reductions'
    fixed order) leave no room for a timing-dependent result.  128 frames (one conv range, twelve-wave attention backward), 64 frames
    (two conv ranges, three-wave workgroups), 6 frames (four ranges, partial tiles everywhere).  Differences are counted on the device:
    one synchronisation at the end."""
    import adafortitran_amd as A
    from adafortitran_amd import synth
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    torch.manual_seed(12)
    model = A.AdaFortiTranEstimator(sc, A.ModelConfig(model_type="adafortitran", patch_size=(3, 2), num_layers=6, model_dim=128, num_head=4,
                                                      device="cuda", dropout=0.1, channel_adaptivity_hidden_sizes=[7, 42, 560],
                                                      adaptive_token_length=6)).train()
    inp = synth.make_inputs(batch, seed=13)
    pil, tgt = torch.from_numpy(inp["pilots"]).cuda(), torch.from_numpy(inp["target"]).cuda()
    meta = tuple(m.cuda() if hasattr(m, "cuda") else m for m in synth.meta_tuple(inp))
    params = list(model.parameters())

    def run():
        for p in params:
            p.grad = None
        torch.manual_seed(99)
        loss = torch.view_as_real(model(pil, meta) - tgt).pow(2).mean()
        loss.backward()
        return torch.cat([loss.detach().reshape(1)] + [p.grad.reshape(-1) for p in params])

    ref = run().clone()
    assert torch.isfinite(ref).all()
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    for _ in range(reps):
        bad += (run().view(torch.int32) != ref.view(torch.int32)).sum()
    assert int(bad) == 0, int(bad)


def _random_train_specs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        p0, p1 = int(rng.integers(1, 5)), int(rng.integers(1, 4))
        gs, gt = int(rng.integers(4, 20)), int(rng.integers(2, 8))
        if gs * gt < 32 or gs * gt > 300 or gs * p0 > 120:
            continue
        d = int(rng.choice([64, 128, 192, 256]))
        out.append(dict(ofdm=(gs * p0, gt * p1), pilot=(int(rng.integers(2, 9)), int(rng.integers(1, 3))), patch=(p0, p1),
                        d=d, act=str(rng.choice(["gelu", "relu"])), adaptive=bool(rng.integers(0, 2)), batch=int(rng.integers(1, 4))))
    # round 5: head dim 16, a 28-token grid and a model_dim off the tuned set, through the whole model
    out.append(dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), d=128, hd=16, act="gelu", adaptive=True, batch=2))
    out.append(dict(ofdm=(12, 14), pilot=(4, 2), patch=(3, 2), d=128, hd=32, act="gelu", adaptive=True, batch=3))
    out.append(dict(ofdm=(48, 14), pilot=(6, 2), patch=(3, 2), d=192, hd=16, act="relu", adaptive=False, batch=2))
    out.append(dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), d=128, hd=64, act="gelu", adaptive=True, batch=2))
    out.append(dict(ofdm=(24, 14), pilot=(6, 2), patch=(3, 2), d=64, hd=64, act="relu", adaptive=False, batch=3))
    out.append(dict(ofdm=(48, 14), pilot=(6, 2), patch=(3, 2), d=96, hd=32, act="gelu", adaptive=True, batch=2))
    out.append(dict(ofdm=(24, 14), pilot=(6, 2), patch=(3, 2), d=160, hd=16, act="relu", adaptive=False, batch=2))
    # late round 5: heads that start anywhere in a 32-feature block (head dims 24 / 48 / 8)
    out.append(dict(ofdm=(48, 14), pilot=(6, 2), patch=(3, 2), d=96, hd=24, act="gelu", adaptive=True, batch=2))
    out.append(dict(ofdm=(48, 14), pilot=(6, 2), patch=(3, 2), d=192, hd=48, act="gelu", adaptive=True, batch=2))
    out.append(dict(ofdm=(24, 14), pilot=(6, 2), patch=(3, 2), d=128, hd=8, act="relu", adaptive=False, batch=3))
    out.append(dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), d=160, hd=40, act="relu", adaptive=True, batch=2))
    return out


@pytest.mark.parametrize("spec", _random_train_specs(8, 77),
                         ids=lambda s: f"{s['ofdm'][0]}x{s['ofdm'][1]}p{s['patch'][0]}x{s['patch'][1]}d{s['d']}hd{s.get('hd', 32)}{'a' if s['adaptive'] else 'f'}")
def test_random_configurations_training_step_matches_autograd(spec):
    """Whole-model loss.backward() on random valid configurations: HIP training kernels (encoder, conv stacks,
    dense layers) against PyTorch-ROCm autograd on the same module, dropout 0.  (The gradients of an untrained model's first layers are
    ~1e-7 and ill-conditioned: tools/debug/hd48_grad_noise.py shows either fp32 implementation 1e-4 .. 3e-2 from float64 by the
    configuration -- the specs below are ones where both sit inside 2e-3; the tight per-layer bounds are in
    test_layer_forward_backward_matches_autograd.)"""
    import adafortitran_amd as A
    from adafortitran_amd import synth, training
    tokens = (spec["ofdm"][0] // spec["patch"][0]) * (spec["ofdm"][1] // spec["patch"][1])
    sc = A.SystemConfig(ofdm=dict(num_scs=spec["ofdm"][0], num_symbols=spec["ofdm"][1]),
                        pilot=dict(num_scs=spec["pilot"][0], num_symbols=spec["pilot"][1]))
    kw = dict(model_type="adafortitran" if spec["adaptive"] else "fortitran", patch_size=spec["patch"], num_layers=2,
              model_dim=spec["d"], num_head=spec["d"] // spec.get("hd", 32), activation=spec["act"], max_seq_len=512,
              pos_encoding_type="learnable", device="cuda", dropout=0.0)
    if spec["adaptive"]:
        kw.update(channel_adaptivity_hidden_sizes=[5, 9, 2 * tokens], adaptive_token_length=6)
    torch.manual_seed(0)
    model = (A.AdaFortiTranEstimator if spec["adaptive"] else A.FortiTranEstimator)(sc, A.ModelConfig(**kw)).train()
    assert all(v is None for v in model.training_backends().values()), model.training_backends()    # every block on the library's kernels
    B = spec["batch"]
    inp = synth.make_inputs(B, ofdm=spec["ofdm"], pilot=spec["pilot"], seed=9)
    pil, tgt = torch.from_numpy(inp["pilots"]).cuda(), torch.from_numpy(inp["target"]).cuda()
    meta = synth.meta_tuple(inp) if spec["adaptive"] else None

    def step(hip):
        model.transformer_encoder.hip_training = hip
        model.initial_enhancer.hip_training = model.final_refiner.hip_training = hip
        training.HipLinear.default_hip_training = hip
        if hasattr(model, "channel_adapter"):
            model.channel_adapter.hip_training = hip
        model.zero_grad()
        out = model(pil, meta) if meta is not None else model(pil)
        loss = torch.nn.functional.mse_loss(torch.view_as_real(out), torch.view_as_real(tgt))
        loss.backward()
        return float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters()}

    try:
        l0, g0 = step(False)
        l1, g1 = step(True)
    finally:
        training.HipLinear.default_hip_training = True
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    for n in g0:
        # the adapter's gradients are ill-conditioned fp32 sums of raw channel conditions (values up to 1400): two
        # PyTorch runs with different reduction orders differ by ~1e-3 of |g|max on them (see test_train_golden.py)
        assert _rel(g1[n], g0[n]) <= (2e-2 if n.startswith("channel_adapter") else 2e-3), n


@pytest.mark.parametrize("hidden,frames", [((7, 42, 560), 12), ((5, 9, 80), 3), ((16, 64, 256), 7)])
def test_channel_adapter_forward_backward_matches_autograd(hidden, frames):
    import adafortitran_amd.blocks as blocks
    from adafortitran_amd import training
    torch.manual_seed(hidden[1])
    ad = blocks.ChannelAdapter(hidden).cuda()
    snr = torch.randint(0, 7, (frames, 1), device="cuda").float() * 5
    ds = torch.randint(1, 8, (frames, 1), device="cuda").float() * 50
    dop = torch.randint(1, 8, (frames, 1), device="cuda").float() * 200
    gy = torch.randn(frames, hidden[2] // 2, 6, device="cuda")

    def run(hip):
        ad.hip_training = hip
        training.HipLinear.default_hip_training = False      # reference path: plain PyTorch-ROCm autograd
        ad.zero_grad()
        y = ad(snr, ds, dop)
        y.backward(gy)
        return y.detach().clone(), [p.grad.clone() for p in ad.parameters()]

    try:
        y0, g0 = run(False)
        y1, g1 = run(True)
    finally:
        training.HipLinear.default_hip_training = True
    assert _rel(y1, y0) <= 1e-5
    for (name, _), a, b in zip(ad.named_parameters(), g1, g0):
        assert _rel(a, b) <= 1e-4, name


def test_grad_scaler_branch_of_the_reference_trainer():
    """The reference's mixed-precision branch (trainer.py:207-217): forward outside autocast, loss under autocast,
    scaler.scale(loss).backward(), unscale_, clip, scaler.step, scaler.update -- on the HIP training kernels, with the
    flat optimizer; same parameters afterwards as the plain fp32 step on PyTorch-ROCm autograd."""
    from adafortitran_amd import synth, training
    from adafortitran_amd.optim import ShardedFlatAdam
    inp = synth.make_inputs(4, seed=21)
    pil, tgt = torch.from_numpy(inp["pilots"]).cuda(), torch.from_numpy(inp["target"]).cuda()
    cat = lambda z: torch.cat((torch.real(z), torch.imag(z)), dim=1)  # noqa: E731

    def run(hip):
        torch.manual_seed(5)
        model = _model("fortitran", 0.0).train()
        model.transformer_encoder.hip_training = hip
        model.initial_enhancer.hip_training = model.final_refiner.hip_training = hip
        training.HipLinear.default_hip_training = hip
        opt = ShardedFlatAdam(model.parameters(), lr=1e-3) if hip else torch.optim.Adam(model.parameters(), lr=1e-3)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
        for _ in range(3):
            opt.zero_grad()
            out = model(pil)
            with torch.autocast("cuda"):
                loss = torch.nn.MSELoss()(cat(out), cat(tgt))
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            scaler.step(opt)
            scaler.update()
        return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()

    try:
        ref = run(False)
        got = run(True)
    finally:
        training.HipLinear.default_hip_training = True
    assert torch.isfinite(got).all()
    diff = (got - ref).abs()
    # Adam divides by sqrt(v): an element whose gradient is ~0 can step +-lr on fp32 noise, so the bound on any single
    # element is steps * lr; the bulk must agree closely
    assert float(diff.max()) <= 3.1e-3 and float(diff.median()) <= 1e-6 and float((diff > 1e-4).float().mean()) <= 0.01


@pytest.mark.parametrize("p,batch", [(0.0, 8), (0.1, 8), (0.0, 128)])
def test_fused_forward_chain_reproduces_the_launch_sequence_tape(switches, p, batch):
    """the tight A/B that backs the training path's golden tolerances.  The fused row-local forward
    (chain_fwd_train_kernel: out_proj + LN1 + FFN + LN2, tape written from its epilogues) against the launch sequence it
    replaced (AFT_TRAIN_UNFUSED_FWD, read per call), same layer, same inputs, same dropout seed: the layer output and EVERY
    tape tensor within 1e-6 of the tensor's max (5e-7 observed: Chan-merged vs two-pass LayerNorm partials, other
    summation orders) -- the masks are the same function of (seed, row, column), so dropout changes nothing here."""
    import ctypes as C
    from adafortitran_amd import _lib
    from adafortitran_amd.training import _layer_struct, layer_params
    lib = _lib.load()
    d, heads = 128, 4
    cfg = _cfg(d, heads)
    layer = _layer(d, heads, "gelu", p).train()
    params = [q.detach().contiguous() for q in layer_params(layer)]
    planes = 2 * batch
    rows = planes * cfg.tokens
    torch.manual_seed(2)
    x = torch.randn(planes, cfg.tokens, d, device="cuda")
    nt, nscr = lib.aft_encoder_tape_bytes(C.byref(cfg), batch), lib.aft_encoder_train_scratch_bytes(C.byref(cfg), batch)
    w = _layer_struct(_abi.AftLayerWeights, params)
    al = lambda n: (n + 63) // 64 * 64  # noqa: E731
    layout = (("qkv", rows * 3 * d), ("attn", rows * d), ("lse", rows * heads), ("s1", rows * d), ("st1", rows * 2), ("x1", rows * d),
              ("a", rows * 2 * d), ("hd", rows * 2 * d), ("s2", rows * d), ("st2", rows * 2))
    res = {}
    for mode in ("fused", "unfused"):
        if mode == "unfused":
            switches.set("AFT_TRAIN_UNFUSED_FWD", "1")
        else:
            switches.unset("AFT_TRAIN_UNFUSED_FWD")
        tape = torch.zeros(nt, dtype=torch.uint8, device="cuda")
        scr = torch.zeros(nscr, dtype=torch.uint8, device="cuda")
        out = torch.empty_like(x)
        _lib.check(lib.aft_encoder_layer_fwd_train_f32(C.byref(cfg), C.byref(w), x.data_ptr(), out.data_ptr(), tape.data_ptr(), nt,
                                                       scr.data_ptr(), nscr, batch, p, 5, _lib.current_stream_ptr(x.device)))
        torch.cuda.synchronize()
        f, off, segs = tape.view(torch.float32), 0, {}
        for name, n in layout:
            segs[name] = f[off:off + n].clone()
            off += al(n)
        segs["out"] = out.view(-1).clone()
        res[mode] = segs
    switches.unset("AFT_TRAIN_UNFUSED_FWD")
    for k in res["fused"]:
        assert _rel(res["fused"][k], res["unfused"][k]) <= 1e-6, k


@pytest.mark.parametrize("grid,planes,p", [((24, 14), 2, 0.0), ((120, 14), 2, 0.1), ((120, 14), 16, 0.1)])
def test_fused_backward_chain_reproduces_the_launch_sequence_gradients(switches, grid, planes, p):
    """... and the fused row-local backward (chain_bwd_kernel) against the five launches it replaced
    (AFT_TRAIN_UNFUSED_BWD): dx and every parameter gradient within 2e-6 of the tensor's max, with and without dropout."""
    from adafortitran_amd.training import HipEncoderLayerFunction, layer_params
    d, heads = 128, 4
    cfg = _cfg(d, heads, grid)
    layer = _layer(d, heads, "gelu", p).train()
    torch.manual_seed(4)
    x0 = torch.randn(planes, cfg.tokens, d, device="cuda")
    gout = torch.randn_like(x0)

    def run(unfused):
        if unfused:
            switches.set("AFT_TRAIN_UNFUSED_BWD", "1")
        else:
            switches.unset("AFT_TRAIN_UNFUSED_BWD")
        layer.zero_grad()
        x = x0.clone().requires_grad_(True)
        out = HipEncoderLayerFunction.apply(x, cfg, p, 5, *layer_params(layer))
        out.backward(gout)
        return [x.grad.clone()] + [q.grad.clone() for q in layer_params(layer)]

    fused, unfused = run(False), run(True)
    switches.unset("AFT_TRAIN_UNFUSED_BWD")
    for name, u, v in zip(["dx", *_abi.LAYER_PARAM_NAMES], fused, unfused):
        assert _rel(u, v) <= 2e-6, name
