"""ctypes wrapper around oracle/libaft_oracle.so (CPU restatement, TEST INFRASTRUCTURE).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package never does.  Parity is pinned against golden
vectors produced by the imported reference (tests/golden/make_golden.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

from adafortitran_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libaft_oracle.so")
_lib = None


class OracleDump(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "upsampled", "conv_enhanced", "tokens6", "embed_in", "x0", "layer_out", "enc_out", "residual")]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "aft_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "adafortitran_amd.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "libaft_oracle.so"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.aft_oracle_forward_f32.restype = C.c_int
        _lib.aft_oracle_adapter_f32.restype = C.c_int
        _lib.aft_oracle_encoder_layer_f32.restype = C.c_int
        _lib.aft_oracle_linear_forward_f32.restype = C.c_int
        _lib.aft_oracle_mse_partial_f32.restype = C.c_int
        _lib.aft_oracle_num_threads.restype = C.c_int
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    """Holds host copies of a state_dict (numpy float32) and runs the C restatement."""

    def __init__(self, cfg: _abi.AftConfig, state: Dict[str, np.ndarray]):
        self.cfg = cfg
        self._keep = {k: _f32(v) for k, v in state.items()}
        self.weights = _abi.make_weights(cfg, lambda k: self._keep[k].ctypes.data,
                                         pos_key=_abi.pos_key_of(self._keep))

    @property
    def tokens(self) -> int:
        return self.cfg.tokens

    def forward(self, pilots: np.ndarray, snr=None, ds=None, dop=None, dump: bool = False):
        """pilots complex64 [B,Ps,Pt] -> complex64 [B,S,T] (+ dict of intermediates)."""
        c = self.cfg
        pil = np.ascontiguousarray(pilots, dtype=np.complex64)
        B = pil.shape[0]
        out = np.empty((B, c.num_scs, c.num_symbols), dtype=np.complex64)
        meta = [None if a is None else _f32(np.asarray(a).reshape(-1)) for a in (snr, ds, dop)]
        d_struct, bufs = None, {}
        if dump:
            S, T, tok, d, L = c.num_scs, c.num_symbols, c.tokens, c.model_dim, c.num_layers
            p = c.patch_scs * c.patch_symbols
            din = p + (6 if c.adaptive else 0)
            bufs = {
                "upsampled": np.zeros((2 * B, S, T), np.float32),
                "conv_enhanced": np.zeros((2 * B, S, T), np.float32),
                "tokens6": np.zeros((B, tok, 6), np.float32),
                "embed_in": np.zeros((2 * B, tok, din), np.float32),
                "x0": np.zeros((2 * B, tok, d), np.float32),
                "layer_out": np.zeros((L, 2 * B, tok, d), np.float32),
                "enc_out": np.zeros((2 * B, tok, p), np.float32),
                "residual": np.zeros((2 * B, S, T), np.float32),
            }
            d_struct = OracleDump(**{k: v.ctypes.data for k, v in bufs.items()})
        rc = lib().aft_oracle_forward_f32(
            C.byref(c), C.byref(self.weights), _p(pil.view(np.float32)), _p(meta[0]), _p(meta[1]),
            _p(meta[2]), _p(out.view(np.float32)), C.c_int(B),
            C.byref(d_struct) if d_struct is not None else None)
        if rc == _abi.AFT_ERR_ARG or rc == _abi.AFT_ERR_SHAPE:
            raise ValueError(f"oracle rejected the call (code {rc})")
        return (out, bufs) if dump else out

    def adapter(self, snr, ds, dop) -> np.ndarray:
        s, d_, p_ = (_f32(np.asarray(a).reshape(-1)) for a in (snr, ds, dop))
        B = s.shape[0]
        out = np.empty((B, self.cfg.tokens, 6), np.float32)
        rc = lib().aft_oracle_adapter_f32(C.byref(self.cfg), C.byref(self.weights), _p(s), _p(d_), _p(p_),
                                          _p(out), C.c_int(B))
        if rc:
            raise ValueError(f"oracle adapter failed (code {rc})")
        return out

    def encoder_layer(self, layer: int, x: np.ndarray) -> np.ndarray:
        """x float32 [planes, tokens, d] -> same shape (planes must be even)."""
        y = _f32(x).copy()
        rc = lib().aft_oracle_encoder_layer_f32(C.byref(self.cfg), C.byref(self.weights), C.c_int(layer),
                                                _p(y), C.c_int(y.shape[0] // 2))
        if rc:
            raise ValueError(f"oracle encoder layer failed (code {rc})")
        return y


def linear_forward(weight: np.ndarray, bias: np.ndarray, pilots: np.ndarray, ofdm) -> np.ndarray:
    w, b = _f32(weight), _f32(bias)
    pil = np.ascontiguousarray(pilots, dtype=np.complex64)
    B = pil.shape[0]
    out = np.empty((B, ofdm[0], ofdm[1]), np.complex64)
    rc = lib().aft_oracle_linear_forward_f32(_p(w), _p(b), _p(pil.view(np.float32)), _p(out.view(np.float32)),
                                             C.c_int(B), C.c_int(w.shape[1]), C.c_int(w.shape[0]))
    if rc:
        raise ValueError(f"oracle linear failed (code {rc})")
    return out


def mse_sum(est: np.ndarray, ref: np.ndarray) -> float:
    e = np.ascontiguousarray(est, dtype=np.complex64)
    r = np.ascontiguousarray(ref, dtype=np.complex64)
    acc = C.c_double(0.0)
    rc = lib().aft_oracle_mse_partial_f32(_p(e.view(np.float32)), _p(r.view(np.float32)), C.byref(acc),
                                          C.c_longlong(e.size))
    if rc:
        raise ValueError("oracle mse failed")
    return acc.value


def num_threads() -> int:
    return int(lib().aft_oracle_num_threads())


def set_threads(n: int) -> None:
    lib().aft_oracle_set_threads(C.c_int(n))
