"""ctypes binding of csrc/libaft_hip.so -- the stub a maintainer of the reference would add
to call the C ABI of include/adafortitran_amd.h from Python.

There is NO fallback: if the shared library is missing or a symbol is absent this module
raises, and every caller on a HIP device fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # imported first on purpose: binds libamdhip64.so.7 to the runtime torch already uses

from . import _abi

# AFT_LIB_PATH: explicit override for A/B-ing kernel variants (tools/); default = the in-tree build
_LIB_PATH = os.environ.get("AFT_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                                                           "libaft_hip.so")
_lib = None


class AftError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load the library once and type its entry points."""
    global _lib
    if _lib is None:
        _lib = load_path(_LIB_PATH)
    return _lib


def load_path(path: str):
    """Load and type one build of the library (the product uses exactly one, `load()`; tools/ab_kernels.py loads
    several variants side by side to time them in one process)."""
    if not os.path.exists(path):
        raise AftError(
            f"{path} is missing: build it with `python -m adafortitran_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for the HIP path.")
    lib = C.CDLL(path)
    for name in _abi.EXPORTED_SYMBOLS:
        if not hasattr(lib, name):
            raise AftError(f"{path} does not export {name}")
    lib.aft_version.restype = C.c_int
    lib.aft_max_batch.restype = C.c_int
    lib.aft_last_error.restype = C.c_char_p
    lib.aft_workspace_bytes.restype = C.c_size_t
    lib.aft_workspace_bytes.argtypes = [C.POINTER(_abi.AftConfig), C.c_int]
    vp, cfgp, wp = C.c_void_p, C.POINTER(_abi.AftConfig), C.POINTER(_abi.AftWeights)
    lib.aft_check_config.argtypes = [cfgp]
    lib.aft_engine_of.argtypes = [cfgp]
    lib.aft_set_switch.argtypes = [C.c_char_p, C.c_char_p]
    lib.aft_get_switch.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    lib.aft_workspace_region.argtypes = [cfgp, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.aft_workspace_lanes.restype = C.c_int
    lib.aft_workspace_lanes.argtypes = [cfgp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    lib.aft_max_batch.argtypes = [cfgp]
    lib.aft_forward_f32.argtypes = [cfgp, wp, vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_int, vp]
    lib.aft_packed_weights_bytes.restype = C.c_size_t
    lib.aft_packed_weights_bytes.argtypes = [cfgp]
    lib.aft_pack_weights_f32.argtypes = [cfgp, wp, vp, C.c_size_t, vp]
    lib.aft_forward_prepacked_f32.argtypes = [cfgp, wp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_int, vp]
    lib.aft_linear_forward_f32.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.aft_mse_partial_f32.argtypes = [vp, vp, vp, C.c_longlong, vp]
    lib.aft_stage_upsample_f32.argtypes = [cfgp, wp, vp, vp, C.c_int, vp]
    lib.aft_stage_adapter_f32.argtypes = [cfgp, wp, vp, vp, vp, vp, C.c_int, vp]
    lib.aft_stage_embed_f32.argtypes = [cfgp, wp, vp, vp, vp, C.c_int, vp]
    lib.aft_stage_encoder_layer_f32.argtypes = [cfgp, wp, C.c_int, vp, vp, C.c_size_t, C.c_int, vp]
    lib.aft_stage_tail_f32.argtypes = [cfgp, wp, vp, vp, vp, C.c_int, vp]
    lib.aft_profile_kernel_f32.argtypes = [cfgp, wp, C.c_int, vp, vp, C.c_size_t, C.c_int, C.c_int, vp]
    lib.aft_debug_fill_lds_f32.argtypes = [C.c_float, vp]
    lib.aft_debug_peek_lds_f32.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.aft_pilot_gather_f32.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.aft_ls_mse_db_f32.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
    lwp, lgp = C.POINTER(_abi.AftLayerWeights), C.POINTER(_abi.AftLayerGrads)
    for name in ("aft_encoder_tape_bytes", "aft_encoder_train_scratch_bytes"):
        getattr(lib, name).restype = C.c_size_t
        getattr(lib, name).argtypes = [cfgp, C.c_int]
    lib.aft_encoder_layer_fwd_train_f32.argtypes = [cfgp, lwp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int,
                                                    C.c_float, C.c_uint64, vp]
    lib.aft_encoder_layer_fwd_train_chained_f32.argtypes = [cfgp, lwp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int,
                                                            C.c_float, C.c_uint64, C.c_int, lwp, vp, C.c_size_t, C.POINTER(C.c_int), vp]
    lib.aft_encoder_layer_bwd_f32.argtypes = [cfgp, lwp, vp, vp, C.c_size_t, vp, vp, lgp, C.c_int, vp, C.c_size_t,
                                              C.c_int, C.c_float, C.c_uint64, vp]
    p4 = C.c_void_p * 4
    lib.aft_conv_enhancer_scratch_bytes.restype = C.c_size_t
    lib.aft_conv_enhancer_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.aft_conv_enhancer_fwd_scratch_bytes.restype = C.c_size_t
    lib.aft_conv_enhancer_fwd_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.aft_conv_enhancer_fwd_train_f32.argtypes = [C.POINTER(p4), C.POINTER(p4), vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp]
    lib.aft_conv_enhancer_bwd_f32.argtypes = [C.POINTER(p4), vp, vp, vp, vp, vp, vp, C.POINTER(p4), C.POINTER(p4), C.c_int, vp,
                                              C.c_size_t, C.c_int, C.c_int, C.c_int, vp]
    lib.aft_dense_bwd_scratch_bytes.restype = C.c_size_t
    lib.aft_dense_bwd_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.aft_dense_fwd_f32.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.aft_dense_bwd_f32.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp]
    p3, p9, i3 = C.c_void_p * 3, C.c_void_p * 9, C.c_int32 * 3
    lib.aft_adapter_fwd_train_f32.argtypes = [C.POINTER(p3), C.POINTER(p9), C.POINTER(p9), C.POINTER(i3), C.c_int, C.c_int, vp, vp, vp, vp]
    lib.aft_adapter_bwd_f32.argtypes = [C.POINTER(p3), C.POINTER(p9), C.POINTER(p9), C.POINTER(i3), C.c_int, C.c_int, vp, vp, vp,
                                        vp, vp, C.POINTER(p9), C.POINTER(p9), C.c_int, vp]
    i6 = [C.c_int] * 6   # planes, num_scs, num_symbols, patch_scs, patch_symbols, model_dim
    lib.aft_embed_bwd_scratch_bytes.restype = C.c_size_t
    lib.aft_embed_bwd_scratch_bytes.argtypes = i6 + [C.c_int]
    lib.aft_embed_fwd_train_f32.argtypes = [vp] * 6 + i6 + [vp]
    lib.aft_embed_bwd_f32.argtypes = [vp] * 9 + [C.c_int, vp, C.c_size_t] + i6 + [vp]
    lib.aft_tail_bwd_scratch_bytes.restype = C.c_size_t
    lib.aft_tail_bwd_scratch_bytes.argtypes = i6
    lib.aft_tail_fwd_train_f32.argtypes = [vp] * 5 + i6 + [vp]
    lib.aft_tail_bwd_f32.argtypes = [vp] * 6 + [C.c_int, vp, C.c_size_t] + i6 + [vp]
    lib.aft_adam_step_f32.argtypes = [vp, vp, vp, vp, C.c_size_t] + [C.c_float] * 6 + [C.c_int, vp]
    for name in _abi.EXPORTED_SYMBOLS:
        if name not in _abi.SIZE_SYMBOLS + ("aft_version", "aft_last_error", "aft_max_batch"):
            getattr(lib, name).restype = C.c_int
    if lib.aft_version() != _abi.AFT_ABI_VERSION:
        raise AftError(f"ABI mismatch: library {lib.aft_version()} vs binding {_abi.AFT_ABI_VERSION}")
    return lib


def check(rc: int) -> None:
    """0 -> ok; argument/shape codes -> ValueError (the reference's convention for bad input,
    fortitran.py:157-158, linear.py:79-83); HIP failures -> RuntimeError."""
    if rc == _abi.AFT_OK:
        return
    msg = load().aft_last_error().decode(errors="replace")
    if rc in (_abi.AFT_ERR_ARG, _abi.AFT_ERR_SHAPE):
        raise ValueError(msg)
    raise AftError(msg)


def set_switch(name: str, value) -> None:
    """A measurement / A-B switch of the library (header: aft_set_switch).  The library reads the AFT_* environment once, when it is
    loaded; afterwards switches change only through this call (``value`` None = unset) -- never through os.environ."""
    check(load().aft_set_switch(name.encode(), None if value is None else str(value).encode()))


def get_switch(name: str):
    """Current value of a switch (str) or None when it is unset."""
    buf = C.create_string_buffer(256)
    return buf.value.decode() if load().aft_get_switch(name.encode(), buf, 256) else None


class switch:
    """``with _lib.switch("AFT_LANES", 1): ...`` -- set a switch for a block and restore what it was."""

    def __init__(self, name: str, value) -> None:
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_switch(self.name)
        set_switch(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_switch(self.name, self.old)
        return False


def current_stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)
