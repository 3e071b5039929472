"""Thin Python layer over the C ABI: pointer tables from torch tensors, workspace cache,
per-stage calls.  PyTorch is plumbing here (device memory + streams); the arithmetic is in
csrc/*.hip.  Nothing in this module falls back to PyTorch math.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _abi, _lib


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev_f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


def config_coverage(cfg: _abi.AftConfig) -> Optional[str]:
    """None when the gfx950 kernels cover ``cfg``, else the library's reason (``aft_check_config``)."""
    lib = _lib.load()
    if lib.aft_check_config(C.byref(cfg)) == _abi.AFT_OK:
        return None
    return lib.aft_last_error().decode(errors="replace")


def conv_enhancer_covered(num_scs: int, num_symbols: int) -> bool:
    """True when the fused conv-stack kernel has an LDS band plan for this grid (training path)."""
    return _lib.load().aft_conv_enhancer_scratch_bytes(1, int(num_scs), int(num_symbols)) > 0


class HipEngine:
    """Owns an ``aft_config``, the weight pointer table and a workspace for one model.

    ``tensors`` maps reference ``state_dict`` keys to float32 CUDA
    tensors; the engine keeps references so the device pointers stay valid.
    """

    def __init__(self, cfg: _abi.AftConfig, tensors: Dict[str, torch.Tensor], lib=None):
        self.lib = lib if lib is not None else _lib.load()
        self.cfg = cfg
        self.device = next(iter(tensors.values())).device
        if self.device.type != "cuda":
            raise ValueError("HipEngine needs tensors on a HIP ('cuda') device")
        self._keep = {}
        for k, v in tensors.items():
            if not v.is_floating_point():
                continue
            if v.dtype != torch.float32 or not v.is_contiguous():
                raise ValueError(f"{k}: HIP path needs contiguous float32 (got {v.dtype})")
            self._keep[k] = v
        self.weights = _abi.make_weights(cfg, lambda k: self._keep[k].data_ptr(), pos_key=_abi.pos_key_of(self._keep))
        # tuning knob for A/B runs (tools/ab_kernels.py, bench.py): AFT_ENCODER_PATH=launches|plane overrides the automatic
        # choice between the layer-by-layer launches and the plane-resident encoder kernel (same output bits either way)
        forced = os.environ.get("AFT_ENCODER_PATH", "")
        if forced:
            cfg.encoder_path = {"auto": _abi.AFT_ENCODER_AUTO, "launches": _abi.AFT_ENCODER_LAUNCHES,
                                "plane": _abi.AFT_ENCODER_PLANE}[forced]
        # scratch is PER STREAM: forwards in flight on different streams must not share the conv_enhanced / x / q / k / v^T
        # regions (nothing orders them against each other); each buffer is allocated with its stream current, so torch's
        # caching allocator re-uses it only in that stream's order
        self._ws: Dict[int, torch.Tensor] = {}
        self._ws_batch: Dict[int, int] = {}
        self.max_batch = int(self.lib.aft_max_batch(C.byref(cfg)))   # 32-bit buffer offsets: larger batches run in chunks
        # OPT-IN fragment-packed image of the encoder's GEMM weights (forward(cache_packed=True): for callers that know
        # their weights are constant, e.g. A/B tools); the default forward is stateless and re-packs inside the call
        lp = f"{_abi._TE}.transformer.layers."
        self._gemm_weights = [v for k, v in self._keep.items() if k.startswith(lp) and k.endswith(
            ("in_proj_weight", "out_proj.weight", "linear1.weight", "linear2.weight"))]
        self._packed: Dict[int, torch.Tensor] = {}
        self._packed_key: Dict[int, tuple] = {}

    MAX_STREAMS = 8      # scratch buffers kept (one per stream that ran a forward); the least recently used is dropped beyond that

    # -- helpers ---------------------------------------------------------------------------
    @property
    def tokens(self) -> int:
        return self.cfg.tokens

    def signature(self):
        return tuple(v.data_ptr() for v in self._keep.values())

    def workspace(self, batch: int) -> torch.Tensor:
        """The calling stream's scratch buffer (grown on demand)."""
        stream = self._stream()
        ws = self._ws.get(stream)
        if ws is not None:   # least-recently-used order: the entry in use moves to the end (eviction below drops the front)
            for table in (self._ws, self._ws_batch, self._packed, self._packed_key):
                if stream in table:
                    table[stream] = table.pop(stream)
        if ws is None or batch > self._ws_batch[stream]:
            nbytes = self.lib.aft_workspace_bytes(C.byref(self.cfg), batch)
            if nbytes == 0:
                _lib.check(self.lib.aft_forward_f32(C.byref(self.cfg), None, None, None, None, None, None, None, 0, batch, None))
                raise ValueError("unsupported configuration")
            if ws is None and len(self._ws) >= self.MAX_STREAMS:
                oldest = next(iter(self._ws))
                for table in (self._ws, self._ws_batch, self._packed, self._packed_key):
                    table.pop(oldest, None)
            with torch.cuda.device(self.device):
                ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[stream], self._ws_batch[stream] = ws, batch
        return ws

    def _stream(self) -> int:
        return _lib.current_stream_ptr(self.device)

    def invalidate_packed(self) -> None:
        self._packed_key.clear()

    def packed_weights(self) -> torch.Tensor:
        """OPT-IN (cache_packed=True): this stream's packed image, re-built (5 us kernel on the current stream) when a GEMM
        weight's autograd version counter moved or after invalidate_packed().  Raw writes through ``.data`` / device
        pointers do NOT move the counters: a caller who makes them must call invalidate_packed() -- which is why the module
        surface does not use the cache."""
        stream = self._stream()
        key = (int(self.cfg.precision), tuple(t._version for t in self._gemm_weights))
        if self._packed_key.get(stream) != key:
            nbytes = self.lib.aft_packed_weights_bytes(C.byref(self.cfg))
            if stream not in self._packed:
                with torch.cuda.device(self.device):
                    self._packed[stream] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.aft_pack_weights_f32(C.byref(self.cfg), C.byref(self.weights), self._packed[stream].data_ptr(),
                                                     nbytes, stream))
            self._packed_key[stream] = key
        return self._packed[stream]

    # -- full forward ----------------------------------------------------------------------
    def forward(self, pilots: torch.Tensor, snr=None, ds=None, dop=None, out: Optional[torch.Tensor] = None,
                cache_packed: bool = False, pinned_inputs: bool = False) -> torch.Tensor:
        """pilots complex64 [B,Ps,Pt] -> complex64 [B,S,T] on the device, asynchronous on the current stream.  Any batch
        size: batches above ``max_batch`` (the ABI's 32-bit-offset limit) run as consecutive chunks, as the reference
        accepts any B (fortitran.py:145-182).  Default = aft_forward_f32, stateless: the encoder weights are re-laid
        into fragment order inside the call.  ``cache_packed``: use this engine's packed image instead
        (aft_forward_prepacked_f32; see packed_weights).  ``pinned_inputs``: pilots / conditions that live on the CPU are
        PINNED host tensors which the kernels read directly (device-addressable; the caller keeps them unchanged until the
        forward has run -- estimators._InputStager); without it every input must be on the engine's device."""
        c = self.cfg
        if pilots.dtype != torch.complex64:
            raise ValueError(f"pilot_symbols must be complex64, got {pilots.dtype}")
        if pilots.dim() != 3 or pilots.shape[1] * pilots.shape[2] != c.pilot_scs * c.pilot_symbols:
            raise ValueError(f"Expected pilot shape (B, {c.pilot_scs}, {c.pilot_symbols}), got {tuple(pilots.shape)}")
        B = pilots.shape[0]
        if pilots.device.type == "cpu" and not pinned_inputs:
            raise ValueError("pilot_symbols must be on the engine's device (or pinned, with pinned_inputs=True)")
        if pilots.device.type == "cpu" and not (pilots.is_pinned() and pilots.is_contiguous()):
            # a pageable (or silently re-copied non-contiguous) host pointer would be a GPU page fault that aborts the process
            raise ValueError("pinned_inputs=True needs contiguous pinned host tensors (pilot_symbols is not)")
        pil = torch.view_as_real(pilots.contiguous())
        metas = [None, None, None]
        if c.adaptive:
            if snr is None or ds is None or dop is None:
                raise ValueError("meta_data is required when channel adaptation is enabled")
            # host conditions are read in place only when they are pinned, contiguous float32; anything else is copied to the device
            metas = [m.reshape(-1) if (pinned_inputs and m.device.type == "cpu" and m.dtype == torch.float32 and m.is_contiguous()
                                       and m.is_pinned())
                     else _dev_f32(m.reshape(-1), self.device) for m in (snr, ds, dop)]
            if any(m.numel() != B for m in metas):
                raise ValueError("meta_data tensors must have one value per frame")
        if out is None:
            out = torch.empty((B, c.num_scs, c.num_symbols), dtype=torch.complex64, device=self.device)
        if B == 0:
            return out
        chunk = min(B, self.max_batch)
        ws = self.workspace(chunk)
        out_r = torch.view_as_real(out)
        packed = self.packed_weights() if cache_packed else None
        for lo in range(0, B, chunk):
            n = min(chunk, B - lo)
            m = [None if t is None else t[lo:lo + n].data_ptr() for t in metas]
            if packed is None:
                rc = self.lib.aft_forward_f32(C.byref(c), C.byref(self.weights), pil[lo:lo + n].data_ptr(), m[0], m[1], m[2],
                                              out_r[lo:lo + n].data_ptr(), ws.data_ptr(), ws.numel(), n, self._stream())
            else:
                rc = self.lib.aft_forward_prepacked_f32(C.byref(c), C.byref(self.weights), packed.data_ptr(),
                                                        pil[lo:lo + n].data_ptr(), m[0], m[1], m[2],
                                                        out_r[lo:lo + n].data_ptr(), ws.data_ptr(), ws.numel(), n,
                                                        self._stream())
            _lib.check(rc)
        return out

    def forward_region(self, name: str, batch: int) -> torch.Tensor:
        """What the last ``forward`` of ``batch`` frames on the current stream left in the workspace (``aft_workspace_region``):
        'conv_enhanced' f32 [2B,S,T], 'tokens6' f32 [B,tokens,6], 'enc_out' f32 [2B,tokens,8|16] -- the production kernels'
        intermediates, for known-answer tests.  A copy, valid whatever runs next."""
        c = self.cfg
        ws = self._ws[self._stream()]
        # the forward may have run as several lanes (contiguous shares of the batch, each with its own slice of the workspace)
        lanes, frames, bases = C.c_int(), (C.c_int * 4)(), (C.c_size_t * 4)()
        _lib.check(self.lib.aft_workspace_lanes(C.byref(c), batch, C.byref(lanes), frames, bases))
        parts = []
        for i in range(lanes.value):
            off, size = C.c_size_t(), C.c_size_t()
            _lib.check(self.lib.aft_workspace_region(C.byref(c), frames[i], _abi.REGION_IDS[name], C.byref(off), C.byref(size)))
            flat = ws[bases[i] + off.value:bases[i] + off.value + size.value].view(torch.float32).clone()
            shape = {"conv_enhanced": (2 * frames[i], c.num_scs, c.num_symbols), "tokens6": (frames[i], self.tokens, 6),
                     "enc_out": (2 * frames[i], self.tokens, -1)}[name]
            parts.append(flat.view(*shape))
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)

    # -- per-stage entry points (tests) ----------------------------------------------------
    def stage_upsample(self, pilots: torch.Tensor) -> torch.Tensor:
        B = pilots.shape[0]
        out = torch.empty((2 * B, self.cfg.num_scs, self.cfg.num_symbols), dtype=torch.float32, device=self.device)
        pil = torch.view_as_real(pilots.contiguous())
        _lib.check(self.lib.aft_stage_upsample_f32(C.byref(self.cfg), C.byref(self.weights), pil.data_ptr(),
                                                   out.data_ptr(), B, self._stream()))
        return out

    def stage_adapter(self, snr, ds, dop) -> torch.Tensor:
        m = [_dev_f32(t.reshape(-1), self.device) for t in (snr, ds, dop)]
        B = m[0].numel()
        out = torch.empty((B, self.tokens, 6), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.aft_stage_adapter_f32(C.byref(self.cfg), C.byref(self.weights), m[0].data_ptr(),
                                                  m[1].data_ptr(), m[2].data_ptr(), out.data_ptr(), B, self._stream()))
        return out

    def stage_embed(self, conv_enhanced: torch.Tensor, tokens6: Optional[torch.Tensor]) -> torch.Tensor:
        B = conv_enhanced.shape[0] // 2
        x = torch.empty((2 * B, self.tokens, self.cfg.model_dim), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.aft_stage_embed_f32(C.byref(self.cfg), C.byref(self.weights), conv_enhanced.data_ptr(),
                                                _ptr(tokens6), x.data_ptr(), B, self._stream()))
        return x

    def stage_encoder_layer(self, layer: int, x: torch.Tensor) -> torch.Tensor:
        """x float32 [2B, tokens, d] -> new tensor, one post-LN encoder layer."""
        y = x.contiguous().clone()
        B = y.shape[0] // 2
        ws = self.workspace(B)
        _lib.check(self.lib.aft_stage_encoder_layer_f32(C.byref(self.cfg), C.byref(self.weights), layer, y.data_ptr(),
                                                        ws.data_ptr(), ws.numel(), B, self._stream()))
        return y

    def stage_tail(self, x: torch.Tensor, conv_enhanced: torch.Tensor) -> torch.Tensor:
        B = conv_enhanced.shape[0] // 2
        out = torch.empty((B, self.cfg.num_scs, self.cfg.num_symbols), dtype=torch.complex64, device=self.device)
        x, conv_enhanced = x.contiguous(), conv_enhanced.contiguous()    # keep both alive across the launch
        _lib.check(self.lib.aft_stage_tail_f32(C.byref(self.cfg), C.byref(self.weights), x.data_ptr(),
                                               conv_enhanced.data_ptr(),
                                               torch.view_as_real(out).data_ptr(), B, self._stream()))
        return out


def fill_lds(value: float, device="cuda:0") -> None:
    """Test hook (aft_debug_fill_lds_f32): every CU's LDS filled with ``value`` on the current stream -- what the next kernels find in
    their LDS at start."""
    dev = torch.device(device)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().aft_debug_fill_lds_f32(C.c_float(value), _lib.current_stream_ptr(dev)))


def peek_lds(workgroups: int = 1024, n: int = 1024, device="cuda:0") -> torch.Tensor:
    """Test hook (aft_debug_peek_lds_f32): what ``workgroups`` fresh workgroups find in the first ``n`` floats of their LDS."""
    dev = torch.device(device)
    with torch.cuda.device(dev):
        out = torch.empty((workgroups, n), dtype=torch.float32, device=dev)
        _lib.check(_lib.load().aft_debug_peek_lds_f32(out.data_ptr(), workgroups, n, _lib.current_stream_ptr(dev)))
    return out


def profile_kernel(eng: HipEngine, which: str, batch: int, reps: int, io: Optional[torch.Tensor] = None) -> None:
    """Enqueue ``reps`` launches of one kernel class on the current stream (bench.py roofline leg)."""
    ws = eng.workspace(batch)
    ptr = None if io is None else (torch.view_as_real(io) if io.is_complex() else io).data_ptr()
    _lib.check(eng.lib.aft_profile_kernel_f32(C.byref(eng.cfg), C.byref(eng.weights), _abi.KERNEL_IDS[which], ptr,
                                              ws.data_ptr(), ws.numel(), batch, reps, eng._stream()))


def engine_from_numpy(cfg: _abi.AftConfig, state: Dict[str, np.ndarray], device="cuda:0", lib=None) -> HipEngine:
    """Upload a numpy state_dict (e.g. from ``synth.make_state_dict``) and build an engine."""
    dev = torch.device(device)
    tensors = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev) for k, v in state.items()}
    return HipEngine(cfg, tensors, lib=lib)


def linear_forward(weight: torch.Tensor, bias: Optional[torch.Tensor], pilots: torch.Tensor, ofdm_size) -> torch.Tensor:
    """LinearEstimator on the HIP device, plane-wise on complex64 pilots [B,Ps,Pt]."""
    lib = _lib.load()
    if pilots.dtype != torch.complex64:
        raise ValueError("pilots must be complex64")
    B = pilots.shape[0]
    out_f, in_f = weight.shape
    if pilots[0].numel() != in_f or ofdm_size[0] * ofdm_size[1] != out_f:
        raise ValueError("shape mismatch between pilots / weight / ofdm_size")
    pil = torch.view_as_real(pilots.contiguous())
    weight = weight.contiguous()
    out = torch.empty((B, ofdm_size[0], ofdm_size[1]), dtype=torch.complex64, device=pilots.device)
    _lib.check(lib.aft_linear_forward_f32(weight.data_ptr(), _ptr(bias), pil.data_ptr(),
                                          torch.view_as_real(out).data_ptr(), B, in_f, out_f,
                                          _lib.current_stream_ptr(pilots.device)))
    return out


def mse_sum(est: torch.Tensor, ref: torch.Tensor, acc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Accumulate sum |est-ref|^2 (float64 device scalar) without a host sync."""
    lib = _lib.load()
    if est.dtype != torch.complex64 or ref.dtype != torch.complex64 or est.shape != ref.shape:
        raise ValueError("est/ref must be complex64 tensors of equal shape")
    if acc is None:
        acc = torch.zeros(1, dtype=torch.float64, device=est.device)
    e, r = torch.view_as_real(est.contiguous()), torch.view_as_real(ref.contiguous())
    _lib.check(lib.aft_mse_partial_f32(e.data_ptr(), r.data_ptr(), acc.data_ptr(), est.numel(),
                                       _lib.current_stream_ptr(est.device)))
    return acc


def pilot_gather(hzero_ls: torch.Tensor, pilot_size, return_counts: bool = False):
    """Sparse LS grid complex64 [B,S,T] (zeros off the pilot positions) -> pilots complex64
    [B,Ps,Pt], the non-zero entries in row-major order (reference dataset.py:116-139).  Raises the
    reference's ValueError when a frame does not hold exactly Ps*Pt non-zero entries -- which costs one
    host sync; ``return_counts=True`` returns ``(pilots, counts int32 [B])`` instead and leaves the check
    (``check_pilot_counts``) to the caller, who can run it off the critical path (ingest.PackedLoader)."""
    lib = _lib.load()
    if hzero_ls.dtype != torch.complex64 or hzero_ls.dim() != 3:
        raise ValueError("hzero_ls must be complex64 [B, S, T]")
    B, n = hzero_ls.shape[0], hzero_ls.shape[1] * hzero_ls.shape[2]
    expected = int(pilot_size[0]) * int(pilot_size[1])
    src = torch.view_as_real(hzero_ls.contiguous())
    out = torch.empty((B, pilot_size[0], pilot_size[1]), dtype=torch.complex64, device=hzero_ls.device)   # the kernel writes every slot
    counts = torch.empty(B, dtype=torch.int32, device=hzero_ls.device)
    _lib.check(lib.aft_pilot_gather_f32(src.data_ptr(), torch.view_as_real(out).data_ptr(), counts.data_ptr(), B, n,
                                        expected, _lib.current_stream_ptr(hzero_ls.device)))
    if return_counts:
        return out, counts
    check_pilot_counts(counts, expected)
    return out


def check_pilot_counts(counts: torch.Tensor, expected: int, first_frame: int = 0) -> None:
    """The reference's "Expected 24 pilot values, got 25" error (dataset.py:133-139) from the kernel's per-frame counts
    (a device or host int32 tensor)."""
    bad = (counts != expected).nonzero()
    if bad.numel():
        i = int(bad[0])
        raise ValueError(f"Expected {expected} pilot values, got {int(counts[i])} (frame {first_frame + i})")


def ls_mse_db(ls: torch.Tensor, ideal: torch.Tensor) -> torch.Tensor:
    """Per-frame LS-baseline MSE in dB, float32 [B] (reference utils.py:248-261 per file)."""
    lib = _lib.load()
    if ls.dtype != torch.complex64 or ideal.dtype != torch.complex64 or ls.shape != ideal.shape or ls.dim() != 3:
        raise ValueError("ls / ideal must be complex64 [B, S, T] of equal shape")
    B, n = ls.shape[0], ls.shape[1] * ls.shape[2]
    db = torch.empty(B, dtype=torch.float32, device=ls.device)
    ls, ideal = ls.contiguous(), ideal.contiguous()      # held until after the launch is enqueued
    _lib.check(lib.aft_ls_mse_db_f32(torch.view_as_real(ls).data_ptr(),
                                     torch.view_as_real(ideal).data_ptr(), db.data_ptr(), B, n,
                                     _lib.current_stream_ptr(ls.device)))
    return db
