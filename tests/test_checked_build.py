"""The CHECKED build of the kernels against the product build (SURVEY.md section 5 "LDS-bounds asserts in debug builds"; VERDICT r5
item 8; GPU AddressSanitizer is not available on the pool).

``libaft_hip_check.so`` = the same sources with -DAFT_CHECKED=1 (``python -m adafortitran_amd.build --variant check -DAFT_CHECKED=1``,
built by ``__graft_entry__.build()``): slot / ring-offset asserts and ring-occupancy tags in the wave-specialised conv pipelines
(k_conv_stream.hip, k_conv_rows.hip), the LDS-flag hand-overs published BEHIND a release fence (the product publishes without one and
relies on gfx950 serving a wave's LDS requests in issue order), polls that trap instead of hanging, workspace-plan invariants in
aft_api.hip.  A violated assert is ``__builtin_trap`` -> the launch faults and the next synchronisation raises.

What is tested: the soak of the conv hand-overs and ten random configurations (both engines) run through the checked build without a
fault, and every output has the product build's BITS -- so the fences the product leaves out change nothing.
"""
import os

import numpy as np
import pytest
import torch

from adafortitran_amd import _abi, _lib, synth
from helpers import DEFAULT_SPEC

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CHECK = os.path.join(os.path.dirname(_lib.lib_path()), "libaft_hip_check.so")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def checked():
    if not os.path.exists(CHECK):
        from adafortitran_amd import build
        build.build_checked()          # hipcc is on the GPU box too; normally the file travels with the tree
    lib = _lib.load_path(CHECK)
    assert lib.aft_version() == _abi.AFT_ABI_VERSION
    return lib


def _pair(cfg, sd, checked):
    from adafortitran_amd.hip_ops import engine_from_numpy
    return engine_from_numpy(cfg, sd, DEV), engine_from_numpy(cfg, sd, DEV, lib=checked)


@pytest.mark.parametrize("batch,reps", [(128, 150), (64, 100), (16, 100), (130, 60)])
def test_conv_hand_over_soak_on_the_checked_build(checked, batch, reps):
    """tests/test_hip_parity.py::test_conv_stream_hand_over_soak through the checked build: one launch sequence, two / four column
    ranges per plane and a remainder launch; no assert fires, no poll runs away, and every output is the product build's bits."""
    hid = (7, 42, 560)
    sd = synth.make_state_dict(**DEFAULT_SPEC, adaptive_hidden=hid, seed=20251114)
    cfg = _abi.make_config(**DEFAULT_SPEC, adaptive_hidden=hid)
    prod, chk = _pair(cfg, sd, checked)
    inp = synth.make_inputs(batch, seed=3)
    pil, meta = _t(inp["pilots"]), [_t(inp[k]) for k in ("snr", "ds", "dop")]
    ref = prod.forward(pil, *meta).clone()
    out = torch.empty_like(ref)
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    for _ in range(reps):
        chk.forward(pil, *meta, out=out)
        bad += (torch.view_as_real(out) != torch.view_as_real(ref)).any().to(torch.int64)
    torch.cuda.synchronize()           # a trapped assert surfaces here
    assert int(bad) == 0


@pytest.mark.parametrize("ofdm,pilot,batch", [((240, 28), (24, 4), 3), ((240, 28), (24, 4), 70), ((180, 20), (12, 4), 2), ((96, 40), (8, 4), 5)])
def test_row_streaming_conv_rings_on_the_checked_build(checked, ofdm, pilot, batch):
    """k_conv_rows.hip (tall planes, config 5's grid): every read of a ring slot asserts the slot holds the symbol it wants (ring tags),
    with 1 / 2 / 4 column ranges per plane; same bits as the product."""
    tokens = (ofdm[0] // 3) * (ofdm[1] // 2)
    spec = dict(ofdm=ofdm, pilot=pilot, patch=(3, 2), num_layers=1, model_dim=64, num_head=2)
    hid = (5, 11, 2 * tokens)
    sd = synth.make_state_dict(**spec, adaptive_hidden=hid, seed=12, max_seq_len=max(512, tokens))
    cfg = _abi.make_config(**spec, adaptive_hidden=hid)
    prod, chk = _pair(cfg, sd, checked)
    inp = synth.make_inputs(batch, ofdm=ofdm, pilot=pilot, seed=13)
    pil, meta = _t(inp["pilots"]), [_t(inp[k]) for k in ("snr", "ds", "dop")]
    want = prod.forward(pil, *meta)
    got = chk.forward(pil, *meta)
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(got), torch.view_as_real(want))


def test_random_configurations_on_the_checked_build(checked):
    """Ten random configurations -- five of the packed engine's, five of the general engine's (tests/test_hip_parity.py's generators) --
    through the checked build: host-side plan invariants hold (lanes, regions), no device assert fires, same bits as the product."""
    from test_hip_parity import _random_specs, _random_specs_general
    for spec in _random_specs(5, 4041) + _random_specs_general(5, 4043):
        tokens = (spec["ofdm"][0] // spec["patch"][0]) * (spec["ofdm"][1] // spec["patch"][1])
        base = dict(ofdm=spec["ofdm"], pilot=spec["pilot"], patch=spec["patch"], num_layers=spec["num_layers"],
                    model_dim=spec["model_dim"], num_head=spec["num_head"])
        hid = (5, 11, 2 * tokens) if spec["adaptive"] else None
        sd = synth.make_state_dict(**base, adaptive_hidden=hid, pos_encoding_type=spec["pos"], max_seq_len=512, seed=7, head_gain=2.0)
        cfg = _abi.make_config(**base, activation=spec["activation"], adaptive_hidden=hid)
        prod, chk = _pair(cfg, sd, checked)
        inp = synth.make_inputs(spec["batch"], ofdm=spec["ofdm"], pilot=spec["pilot"], seed=8)
        meta = [(_t(inp[k]) if spec["adaptive"] else None) for k in ("snr", "ds", "dop")]
        want = prod.forward(_t(inp["pilots"]), *meta)
        got = chk.forward(_t(inp["pilots"]), *meta)
        torch.cuda.synchronize()
        assert torch.equal(torch.view_as_real(got), torch.view_as_real(want)), spec


def test_lanes_plan_invariants_on_the_checked_build(checked):
    """Forwards that run as two lanes (64 / 96 / 129 frames of the default model): the checked build verifies on every call that the
    shares partition the batch and their workspace slices are disjoint and inside the buffer."""
    hid = (7, 42, 560)
    sd = synth.make_state_dict(**DEFAULT_SPEC, adaptive_hidden=hid, seed=5)
    cfg = _abi.make_config(**DEFAULT_SPEC, adaptive_hidden=hid)
    prod, chk = _pair(cfg, sd, checked)
    for batch in (64, 96, 129):
        inp = synth.make_inputs(batch, seed=6)
        pil, meta = _t(inp["pilots"]), [_t(inp[k]) for k in ("snr", "ds", "dop")]
        assert torch.equal(torch.view_as_real(chk.forward(pil, *meta)), torch.view_as_real(prod.forward(pil, *meta)))
    torch.cuda.synchronize()
