"""ctypes mirror of include/adafortitran_amd.h (structs + state_dict -> pointer table).

Kept free of any library loading so that both the product loader (_lib.py) and the
test-side oracle wrapper (oracle/oracle.py) can share the struct definitions.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional

AFT_ABI_VERSION = 8
AFT_ENGINE_PACKED, AFT_ENGINE_GENERAL = 0, 1
AFT_OK, AFT_ERR_ARG, AFT_ERR_SHAPE, AFT_ERR_HIP = 0, 1, 2, 3
AFT_ACT_RELU, AFT_ACT_GELU = 0, 1
AFT_ENCODER_AUTO, AFT_ENCODER_LAUNCHES, AFT_ENCODER_PLANE = 0, 1, 2
AFT_PRECISION_F32, AFT_PRECISION_BF16X3 = 0, 1

_fp = C.c_void_p  # const float* -- kept untyped so torch data_ptr() ints and numpy ptrs both fit


class AftConfig(C.Structure):
    _fields_ = [
        ("num_scs", C.c_int32), ("num_symbols", C.c_int32),
        ("pilot_scs", C.c_int32), ("pilot_symbols", C.c_int32),
        ("patch_scs", C.c_int32), ("patch_symbols", C.c_int32),
        ("num_layers", C.c_int32), ("model_dim", C.c_int32), ("num_head", C.c_int32),
        ("activation", C.c_int32), ("adaptive", C.c_int32),
        ("hidden", C.c_int32 * 3), ("encoder_path", C.c_int32), ("precision", C.c_int32),
    ]

    @property
    def tokens(self) -> int:
        return (self.num_scs // self.patch_scs) * (self.num_symbols // self.patch_symbols)


class AftLayerWeights(C.Structure):
    _fields_ = [(n, _fp) for n in (
        "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b",
        "lin2_w", "lin2_b", "norm1_w", "norm1_b", "norm2_w", "norm2_b")]


LAYER_FIELDS = ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b",
                "lin2_w", "lin2_b", "norm1_w", "norm1_b", "norm2_w", "norm2_b")
#: nn.TransformerEncoderLayer parameter names in LAYER_FIELDS order
LAYER_PARAM_NAMES = ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                     "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                     "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


class AftLayerGrads(C.Structure):
    _fields_ = [(n, _fp) for n in LAYER_FIELDS]


class AftWeights(C.Structure):
    _fields_ = [
        ("up_w", _fp), ("up_b", _fp),
        ("enh_w", _fp * 4), ("enh_b", _fp * 4),
        ("ref_w", _fp * 4), ("ref_b", _fp * 4),
        ("ada_w", (_fp * 3) * 3), ("ada_b", (_fp * 3) * 3),
        ("lin1_w", _fp), ("lin1_b", _fp),
        ("pos", _fp),
        ("lin2_w", _fp), ("lin2_b", _fp),
        ("layers", C.POINTER(AftLayerWeights)),     # host array of num_layers entries (any layer count)
    ]


def make_config(*, ofdm, pilot, patch, num_layers: int, model_dim: int, num_head: int,
                activation: str = "gelu", adaptive_hidden=None) -> AftConfig:
    cfg = AftConfig()
    cfg.num_scs, cfg.num_symbols = int(ofdm[0]), int(ofdm[1])
    cfg.pilot_scs, cfg.pilot_symbols = int(pilot[0]), int(pilot[1])
    cfg.patch_scs, cfg.patch_symbols = int(patch[0]), int(patch[1])
    cfg.num_layers, cfg.model_dim, cfg.num_head = int(num_layers), int(model_dim), int(num_head)
    cfg.activation = AFT_ACT_GELU if activation == "gelu" else AFT_ACT_RELU
    cfg.adaptive = 1 if adaptive_hidden is not None else 0
    for i in range(3):
        cfg.hidden[i] = int(adaptive_hidden[i]) if adaptive_hidden is not None else 0
    return cfg


def config_from_pydantic(system_config, model_config, adaptive: bool) -> AftConfig:
    return make_config(
        ofdm=(system_config.ofdm.num_scs, system_config.ofdm.num_symbols),
        pilot=(system_config.pilot.num_scs, system_config.pilot.num_symbols),
        patch=tuple(model_config.patch_size), num_layers=model_config.num_layers,
        model_dim=model_config.model_dim, num_head=model_config.num_head,
        activation=model_config.activation,
        adaptive_hidden=tuple(model_config.channel_adaptivity_hidden_sizes) if adaptive else None)


_ENC = ("snr_encoder", "ds_encoder", "dop_encoder")
_TE = "transformer_encoder"


def make_weights(cfg: AftConfig, ptr: Callable[[str], int], pos_key: Optional[str] = None) -> AftWeights:
    """Fill the pointer table from reference ``state_dict`` key names
    (SURVEY.md Appendix A).  ``ptr(key)`` returns the address of that tensor's
    contiguous float32 storage (device address for the HIP library, host address
    for the oracle)."""
    w = AftWeights()
    w.up_w, w.up_b = ptr("pilot_upsampler.weight"), ptr("pilot_upsampler.bias")
    for i, slot in enumerate((0, 2, 4, 6)):
        w.enh_w[i] = ptr(f"initial_enhancer.conv_block.{slot}.weight")
        w.enh_b[i] = ptr(f"initial_enhancer.conv_block.{slot}.bias")
        w.ref_w[i] = ptr(f"final_refiner.conv_block.{slot}.weight")
        w.ref_b[i] = ptr(f"final_refiner.conv_block.{slot}.bias")
    if cfg.adaptive:
        for e, enc in enumerate(_ENC):
            for j, slot in enumerate((0, 2, 4)):
                w.ada_w[e][j] = ptr(f"channel_adapter.{enc}.{slot}.weight")
                w.ada_b[e][j] = ptr(f"channel_adapter.{enc}.{slot}.bias")
    w.lin1_w, w.lin1_b = ptr(f"{_TE}.linear_1.weight"), ptr(f"{_TE}.linear_1.bias")
    w.lin2_w, w.lin2_b = ptr(f"{_TE}.linear_2.weight"), ptr(f"{_TE}.linear_2.bias")
    w.pos = ptr(pos_key or f"{_TE}.positional_encoding.position_embeddings")
    table = (AftLayerWeights * cfg.num_layers)()
    w.layers = C.cast(table, C.POINTER(AftLayerWeights))
    w._layer_table = table          # the struct holds a bare pointer: keep the host array alive with it
    for i in range(cfg.num_layers):
        lp = f"{_TE}.transformer.layers.{i}"
        lw = table[i]
        lw.in_proj_w, lw.in_proj_b = ptr(lp + ".self_attn.in_proj_weight"), ptr(lp + ".self_attn.in_proj_bias")
        lw.out_proj_w, lw.out_proj_b = ptr(lp + ".self_attn.out_proj.weight"), ptr(lp + ".self_attn.out_proj.bias")
        lw.lin1_w, lw.lin1_b = ptr(lp + ".linear1.weight"), ptr(lp + ".linear1.bias")
        lw.lin2_w, lw.lin2_b = ptr(lp + ".linear2.weight"), ptr(lp + ".linear2.bias")
        lw.norm1_w, lw.norm1_b = ptr(lp + ".norm1.weight"), ptr(lp + ".norm1.bias")
        lw.norm2_w, lw.norm2_b = ptr(lp + ".norm2.weight"), ptr(lp + ".norm2.bias")
    return w


def pos_key_of(state: Dict[str, object]) -> str:
    k = f"{_TE}.positional_encoding.position_embeddings"
    return k if k in state else f"{_TE}.positional_encoding.pe"


#: every symbol include/adafortitran_amd.h declares (tests check the .so exports them all)
EXPORTED_SYMBOLS = (
    "aft_version", "aft_last_error", "aft_check_config", "aft_engine_of", "aft_set_switch", "aft_get_switch", "aft_max_batch", "aft_workspace_bytes", "aft_workspace_region", "aft_workspace_lanes", "aft_forward_f32",
    "aft_packed_weights_bytes", "aft_pack_weights_f32", "aft_forward_prepacked_f32",
    "aft_linear_forward_f32", "aft_mse_partial_f32", "aft_stage_upsample_f32",
    "aft_stage_adapter_f32", "aft_stage_embed_f32", "aft_stage_encoder_layer_f32",
    "aft_stage_tail_f32", "aft_profile_kernel_f32", "aft_debug_fill_lds_f32", "aft_debug_peek_lds_f32", "aft_pilot_gather_f32", "aft_ls_mse_db_f32",
    "aft_encoder_tape_bytes", "aft_encoder_train_scratch_bytes",
    "aft_encoder_layer_fwd_train_f32", "aft_encoder_layer_fwd_train_chained_f32", "aft_encoder_layer_bwd_f32", "aft_adam_step_f32",
    "aft_conv_enhancer_fwd_train_f32", "aft_conv_enhancer_scratch_bytes", "aft_conv_enhancer_fwd_scratch_bytes", "aft_conv_enhancer_bwd_f32",
    "aft_dense_fwd_f32", "aft_dense_bwd_scratch_bytes", "aft_dense_bwd_f32",
    "aft_adapter_fwd_train_f32", "aft_adapter_bwd_f32",
    "aft_embed_bwd_scratch_bytes", "aft_embed_fwd_train_f32", "aft_embed_bwd_f32",
    "aft_tail_bwd_scratch_bytes", "aft_tail_fwd_train_f32", "aft_tail_bwd_f32",
)
#: size queries (return size_t, not a status code)
SIZE_SYMBOLS = ("aft_workspace_bytes", "aft_packed_weights_bytes", "aft_encoder_tape_bytes", "aft_encoder_train_scratch_bytes",
                "aft_conv_enhancer_scratch_bytes", "aft_conv_enhancer_fwd_scratch_bytes", "aft_dense_bwd_scratch_bytes",
                "aft_embed_bwd_scratch_bytes", "aft_tail_bwd_scratch_bytes")
REGION_IDS = {"conv_enhanced": 0, "tokens6": 1, "enc_out": 2}   # aft_workspace_region
KERNEL_IDS = {"upsample": 0, "embed": 1, "qkv": 2, "attention": 3, "chain": 4, "tail": 5, "chain_last": 6, "encoder_plane": 7, "prologue": 8}
