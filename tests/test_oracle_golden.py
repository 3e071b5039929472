"""Pin the CPU oracle (oracle/aft_oracle.c) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py with the imported reference)."""
import numpy as np
import pytest

from helpers import Golden, TOL_ORACLE_OUT, TOL_ORACLE_STAGE, max_rel

ALL_SETS = ["T_tiny_ada", "T_tiny_forti", "D_forti", "A_ada", "DH_forti_hot", "AH_ada_mid", "AS_ada_sin_relu",
            "H16_ada_heads8", "H64_forti_heads2", "S28_ada_tokens28",     # head dims 16 / 64, a grid of 28 tokens
            "H24_ada_d96_heads4", "H48_forti_d192_heads4",                # heads of 24 / 48 features (not aligned with 32-feature blocks)
            # round 6: the general engine's shapes (model_dim 512, heads of 128 / 56 / 25 features, a 24-element patch) and 40 layers
            "W512_ada_d512_heads8", "H128_forti_d256_heads2", "H56_ada_d224_heads4", "D200_forti_d200_heads8", "P24_ada_patch12x2",
            "L40_forti_layers40"]


@pytest.mark.parametrize("name", ALL_SETS)
def test_oracle_matches_reference_output(oracle_lib, name):
    g = Golden(name)
    orc = oracle_lib.Oracle(g.abi_config(), g.state_dict())
    out = orc.forward(g["pilots"], *g.meta_arrays())
    ref = g["out"]
    assert out.shape == ref.shape and out.dtype == np.complex64
    err = np.abs(out - ref).max()
    assert err <= TOL_ORACLE_OUT * max(1.0, np.abs(ref).max()), err


@pytest.mark.parametrize("name", ["T_tiny_ada", "T_tiny_forti"])
def test_oracle_every_stage_tiny(oracle_lib, name):
    g = Golden(name)
    orc = oracle_lib.Oracle(g.abi_config(), g.state_dict())
    _, dump = orc.forward(g["pilots"], *g.meta_arrays(), dump=True)
    stages = ["upsampled", "conv_enhanced", "embed_in", "x0", "layer_out", "enc_out", "residual"]
    if g.adaptive:
        stages.append("tokens6")
    for st in stages:
        assert dump[st].shape == g[st].shape, st
        assert max_rel(dump[st], g[st]) <= TOL_ORACLE_STAGE, (st, max_rel(dump[st], g[st]))


@pytest.mark.parametrize("name", ["D_forti", "A_ada", "DH_forti_hot", "AH_ada_mid"])
def test_oracle_intermediates_default(oracle_lib, name):
    g = Golden(name)
    orc = oracle_lib.Oracle(g.abi_config(), g.state_dict())
    _, dump = orc.forward(g["pilots"], *g.meta_arrays(), dump=True)
    for st in ("conv_enhanced", "enc_out", "residual", "tokens6"):
        if st in g:
            assert max_rel(dump[st], g[st]) <= TOL_ORACLE_STAGE, (st, max_rel(dump[st], g[st]))
    if "x0_f0" in g:
        assert max_rel(dump["x0"][:2], g["x0_f0"]) <= TOL_ORACLE_STAGE
    if "layer_first_last_p0" in g:
        L = g.spec["num_layers"]
        got = np.stack([dump["layer_out"][0, 0], dump["layer_out"][L - 1, 0]])
        assert max_rel(got, g["layer_first_last_p0"]) <= TOL_ORACLE_STAGE


def test_oracle_config5_large(oracle_lib):
    g = Golden("C5_ada_large")
    orc = oracle_lib.Oracle(g.abi_config(), g.state_dict())
    out = orc.forward(g["pilots"], *g.meta_arrays())
    assert np.abs(out - g["out"]).max() <= TOL_ORACLE_OUT * max(1.0, np.abs(g["out"]).max())


def test_oracle_encoder_layer_entry_matches_forward(oracle_lib):
    g = Golden("T_tiny_ada")
    orc = oracle_lib.Oracle(g.abi_config(), g.state_dict())
    y = orc.encoder_layer(0, g["x0"])
    assert max_rel(y, g["layer_out"][0]) <= TOL_ORACLE_STAGE


def test_oracle_linear_estimator_planewise(oracle_lib):
    from adafortitran_amd import synth
    g = Golden("L_linear")
    seed = g.meta["seed"]
    w = synth.uniform_pm(seed, "linear.weight", (1680, 24), 1 / np.sqrt(24))
    b = synth.uniform_pm(seed, "linear.bias", (1680,), 1 / np.sqrt(24))
    out = oracle_lib.linear_forward(w, b, g["pilots"], (120, 14))
    assert np.abs(out - g["out"]).max() <= TOL_ORACLE_OUT
    assert g.meta["complex_input_error"]  # the reference itself raises on complex input


@pytest.mark.parametrize("name", ["D_forti", "A_ada"])
def test_oracle_metric(oracle_lib, name):
    """2*MSELoss(cat(Re,Im)) == mean |est-ref|^2 over complex elements (utils.py:164-180)."""
    g = Golden(name)
    s = oracle_lib.mse_sum(g["out"], g["target"])
    mse = s / g["out"].size
    assert abs(mse - g.meta["metric_2xmse"]) <= 1e-6 * g.meta["metric_2xmse"]
    assert abs(10 * np.log10(mse) - g.meta["metric_db"]) <= 1e-5


def test_oracle_rejects_missing_meta(oracle_lib):
    g = Golden("T_tiny_ada")
    orc = oracle_lib.Oracle(g.abi_config(), g.state_dict())
    with pytest.raises(ValueError):
        orc.forward(g["pilots"])  # fortitran.py:157-158
