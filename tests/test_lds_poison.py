"""What a kernel finds in its LDS at start is whatever the previous kernel on that CU left there.  A kernel that reads an LDS word it
never wrote -- even only to multiply it with a zero weight -- computes garbage that depends on the launch history: round 6's
``embed_any_kernel`` did exactly that (0 x NaN), and it showed only where other tests' kernels had run before.  The allocator-poisoning
tests cannot reach LDS; ``aft_debug_fill_lds_f32`` can (every CU's LDS filled with one value on the stream).  Every path here is run
clean, then behind a NaN / 1e30 / inf fill, and must produce the same BITS."""
import numpy as np
import pytest
import torch

import adafortitran_amd as A
from adafortitran_amd import _abi, synth
from adafortitran_amd.hip_ops import engine_from_numpy, fill_lds
from helpers import DEFAULT_SPEC

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
POISONS = [float("nan"), 1e30, float("-inf")]


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_the_fill_reaches_what_the_next_kernel_finds():
    """The hook checks itself: behind a fill, fresh workgroups on (nearly) every CU find the value in their LDS -- the LDS is not
    cleared between kernels, which is the whole point."""
    from adafortitran_amd.hip_ops import peek_lds
    for value in (123.5, -7.25):
        fill_lds(value, DEV)
        seen = peek_lds(2048, 2048, DEV)
        frac = float((seen == value).float().mean())
        assert frac > 0.9, (value, frac)
    fill_lds(float("nan"), DEV)
    assert float(torch.isnan(peek_lds(2048, 2048, DEV)).float().mean()) > 0.9


FORWARD_CASES = {
    "default_B8": (dict(DEFAULT_SPEC), (7, 42, 560), 8),
    "default_B128": (dict(DEFAULT_SPEC), (7, 42, 560), 128),
    "default_B40_two_lanes": (dict(DEFAULT_SPEC), (7, 42, 560), 40),
    "forti_d256_h8": (dict(DEFAULT_SPEC, num_layers=2, model_dim=256, num_head=8), None, 9),
    "hd16_d128_h8": (dict(DEFAULT_SPEC, num_layers=2, num_head=8), (7, 42, 560), 5),
    "hd8_d128_h16": (dict(DEFAULT_SPEC, num_layers=2, num_head=16), (7, 42, 560), 5),
    "hd48_d192_h4": (dict(DEFAULT_SPEC, num_layers=2, model_dim=192, num_head=4), None, 5),
    "tokens28": (dict(ofdm=(12, 14), pilot=(4, 2), patch=(3, 2), num_layers=2, model_dim=64, num_head=2), (7, 42, 56), 7),
    "grid_66x12_banded_conv": (dict(ofdm=(66, 12), pilot=(11, 3), patch=(3, 3), num_layers=1, model_dim=64, num_head=2), None, 3),
    "tall_240x28_rows_conv": (dict(ofdm=(240, 28), pilot=(24, 4), patch=(3, 2), num_layers=1, model_dim=64, num_head=2), (5, 11, 2240), 3),
    # the general engine (round 6)
    "general_d512_h8": (dict(DEFAULT_SPEC, num_layers=2, model_dim=512, num_head=8), (7, 42, 560), 5),
    "general_d256_h2_hd128": (dict(DEFAULT_SPEC, num_layers=2, model_dim=256, num_head=2), None, 5),
    "general_d200_h8_hd25": (dict(DEFAULT_SPEC, num_layers=2, model_dim=200, num_head=8), (7, 42, 560), 5),
    "general_patch24": (dict(ofdm=(96, 14), pilot=(12, 2), patch=(12, 2), num_layers=1, model_dim=128, num_head=4), (5, 11, 112), 4),
}


@pytest.mark.parametrize("case", sorted(FORWARD_CASES))
def test_forward_bits_do_not_depend_on_what_the_lds_held(case):
    spec, hid, batch = FORWARD_CASES[case]
    sd = synth.make_state_dict(**spec, adaptive_hidden=hid, seed=11, head_gain=2.0, max_seq_len=2240 if spec["ofdm"][0] == 240 else 512)
    cfg = _abi.make_config(**spec, adaptive_hidden=hid)
    eng = engine_from_numpy(cfg, sd, DEV)
    inp = synth.make_inputs(batch, ofdm=spec["ofdm"], pilot=spec["pilot"], seed=12)
    meta = [(_t(inp[k]) if hid else None) for k in ("snr", "ds", "dop")]
    pil = _t(inp["pilots"])
    ref = eng.forward(pil, *meta).clone()
    assert torch.isfinite(torch.view_as_real(ref)).all()
    for value in POISONS:
        fill_lds(value, DEV)
        out = eng.forward(pil, *meta)
        assert torch.equal(torch.view_as_real(out), torch.view_as_real(ref)), (case, value)


@pytest.mark.parametrize("d,heads,ofdm", [(128, 4, (120, 14)), (256, 8, (48, 14)), (128, 8, (24, 14)), (512, 8, (24, 14)), (256, 2, (48, 14)), (200, 8, (24, 14))])
def test_training_layer_bits_do_not_depend_on_what_the_lds_held(d, heads, ofdm):
    """One encoder layer, forward + backward on the training kernels (fused row-local kernels at 128, launch sequences elsewhere, padded
    and multi-block heads), dropout on: output and every gradient the same bits behind an LDS fill."""
    from adafortitran_amd.training import HipEncoderLayerFunction, layer_params
    cfg = _abi.make_config(ofdm=ofdm, pilot=(4, 2), patch=(3, 2), num_layers=1, model_dim=d, num_head=heads)
    torch.manual_seed(5)
    layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=heads, dim_feedforward=2 * d, activation="gelu", dropout=0.1,
                                             batch_first=True).to(DEV).train()
    x0 = torch.randn(4, cfg.tokens, d, device=DEV)
    gout = torch.randn(4, cfg.tokens, d, device=DEV)

    def run():
        layer.zero_grad()
        x = x0.clone().requires_grad_(True)
        out = HipEncoderLayerFunction.apply(x, cfg, 0.1, 77, *layer_params(layer))
        out.backward(gout)
        return [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer_params(layer)]

    ref = run()
    assert all(torch.isfinite(t).all() for t in ref)
    for value in POISONS[:2]:
        fill_lds(value, DEV)
        got = run()
        assert all(torch.equal(a, b) for a, b in zip(got, ref)), (d, heads, value)


@pytest.mark.parametrize("adaptive", [False, True])
def test_training_step_bits_do_not_depend_on_what_the_lds_held(adaptive):
    """The whole model's training step (conv stacks on the 16x16x4 training kernel, adapter, dense layers, encoder): loss and every
    gradient the same bits behind an LDS fill (dropout 0: the step is a function of its inputs)."""
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    kw = dict(model_type="adafortitran" if adaptive else "fortitran", patch_size=(3, 2), num_layers=2, model_dim=128, num_head=4, device="cuda",
              dropout=0.0)
    if adaptive:
        kw.update(channel_adaptivity_hidden_sizes=[7, 42, 560], adaptive_token_length=6)
    torch.manual_seed(9)
    model = (A.AdaFortiTranEstimator if adaptive else A.FortiTranEstimator)(sc, A.ModelConfig(**kw)).train()
    inp = synth.make_inputs(6, seed=10)
    pil, tgt = torch.from_numpy(inp["pilots"]), torch.from_numpy(inp["target"]).to(DEV)
    meta = synth.meta_tuple(inp) if adaptive else None

    def run():
        model.zero_grad()
        est = model(pil, meta) if adaptive else model(pil)
        loss = torch.view_as_real(est - tgt).pow(2).mean()
        loss.backward()
        return [loss.detach().clone()] + [p.grad.clone() for p in model.parameters()]

    ref = run()
    assert all(torch.isfinite(t).all() for t in ref)
    for value in POISONS[:2]:
        fill_lds(value, DEV)
        got = run()
        assert all(torch.equal(a, b) for a, b in zip(got, ref)), value
