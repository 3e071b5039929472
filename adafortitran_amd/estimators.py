"""Drop-in estimators: the reference's ``src/models`` module surface over the gfx950 C ABI.

Kept byte-compatible with the reference where callers can see it:
constructor ``(system_config, model_config)``, ``forward(pilot_symbols, meta_data=None)``,
``get_model_info()`` keys, public attributes, ``state_dict`` keys/shapes, class identities
(``AdaFortiTranEstimator`` is the trainer's dispatch key, reference trainer.py:185-193),
exceptions (``ValueError`` for missing meta / bad shapes).

Execution paths of ``forward``:
  * HIP device, ``eval()`` and autograd off, float32 parameters  ->  ONE call into
    ``aft_forward_f32`` (hand-written HIP).  A missing/failed extension RAISES here; there
    is no silent PyTorch fallback for this case.
  * HIP device with autograd on (the reference trainer's ``train_epoch``) -> the differentiable
    composite in blocks.py, whose encoder layers, conv stacks, dense layers and adapter MLPs run the
    library's hand-written forward/backward kernels behind ``torch.autograd.Function`` (training.py,
    SURVEY.md 8f-1); only reshapes, concatenations and the loss are PyTorch ops.
  * ``device: cpu`` -> the same composite on ATen's CPU kernels (what the reference itself runs).

Coverage is ONE predicate, asked once at construction (``aft_check_config``).  Covered: any layer count, model_dim
any multiple of 8 up to 512, any num_head that divides it with heads of up to 128 features, patches of up to 32
elements -- the tuned fragment-packed engine where its shape conditions hold (model_dim a multiple of 32 up to 256,
head dims 8 .. 64 in steps of 8 except 56, patches of up to 16 elements), the row-major general engine elsewhere
(``hip_engine_name()`` says which).  A configuration the reference accepts beyond that (a larger model_dim or
head, a grid with no LDS band plan) is refused with a ``ValueError`` when the model is built on a HIP device --
before any training -- instead of failing in the first ``eval()`` forward.  ``AFT_ALLOW_COMPOSITE=1`` in the environment opts into running such a model entirely on the
PyTorch-ROCm composite (logged).  The TRAINING kernels cover what inference covers; where a block of an accepted
model is nevertheless differentiated by PyTorch-ROCm autograd (switched off by hand, a conv grid without a band plan), the
constructor logs a warning naming the block and the reason, and ``training_backends()`` returns the same.
"""
from __future__ import annotations

import logging
import os
from typing import List, Optional, Tuple

import torch
from torch import nn

from . import _abi
from .blocks import (ChannelAdapter, ConvEnhancer, InversePatchEmbedding, PatchEmbedding,
                     TransformerEncoderForChannels)
from .config import ModelConfig, SystemConfig, check_shape_coupling
from .training import HipLinear


class _InputStager:
    """The model-owned H2D transfer of ``forward`` (reference fortitran.py:167-173) on a HIP device.  Pilots + the three
    condition vectors of a batch (25 KB at B=128) are packed into ONE slot of a ring of pinned host buffers.

    * Inference (the HIP engine): **no copy is enqueued at all** -- pinned host memory is device-addressable, so the
      conv head and the adapter kernel read the slot directly over PCIe (``host_views``; the ABI takes any
      device-addressable pointer).  A copy-engine transfer in front of the first launch cost ~80 us per forward (the
      hop between the copy engine and the compute queue), 5 % of the 1.57-ms step; the direct read is hidden under the
      conv head's weight-staging phase.
    * Training (torch ops consume the inputs): ONE asynchronous copy on the caller's current stream into a per-call
      device allocation (``to_device``), instead of four pageable ``.to(device)`` copies that each end in a stream
      synchronisation.

    Safety: a slot is re-used only after the event recorded behind its last reader (``release``: the kernels of the
    forward that read it, or the copy) has completed; the event is recorded on the reader's stream of the MODEL's device
    (which need not be the current device).  Device-side staging buffers are per call (torch's caching allocator re-uses
    a block only in stream order of the stream it was allocated on), so forwards issued from different streams never
    share one."""

    SLOTS = 8

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.nbytes = 0
        self.host, self.events = [], []
        self.turn = 0

    def _resize(self, nbytes: int) -> None:
        for ev in self.events:
            if ev is not None:
                ev.synchronize()                         # readers of the old, smaller ring
        self.nbytes = max(nbytes, 4096)
        self.host = [torch.empty(self.nbytes, dtype=torch.uint8).pin_memory() for _ in range(self.SLOTS)]
        self.events = [None] * self.SLOTS

    def fill(self, pilots: Optional[torch.Tensor], conds: Optional[List[torch.Tensor]]):
        """Copy the CPU inputs into the next ring slot (either part may be None = already on the device).
        Returns (slot, total bytes, pinned pilots view | None, pinned [snr, ds, dop] views | None)."""
        B = pilots.shape[0] if pilots is not None else conds[0].numel()
        pil_bytes = pilots.numel() * 8 if pilots is not None else 0
        off = (pil_bytes + 15) // 16 * 16
        total = off + (3 * B * 4 if conds is not None else 0)
        if total > self.nbytes:
            self._resize(total)
        slot = self.turn
        self.turn = (slot + 1) % self.SLOTS
        if self.events[slot] is not None:
            self.events[slot].synchronize()          # whatever last read this pinned buffer has finished
        host = self.host[slot]
        pil_h = cond_h = None
        if pilots is not None:
            pil_h = host[:pil_bytes].view(torch.complex64).view(pilots.shape)
            pil_h.copy_(pilots)
        if conds is not None:
            cond_h = [host[off + 4 * B * i: off + 4 * B * (i + 1)].view(torch.float32) for i in range(3)]
            for dst, c in zip(cond_h, conds):
                dst.copy_(c.reshape(-1))
        return slot, total, pil_h, cond_h

    def release(self, slot: int) -> None:
        """Record the slot's guard behind everything enqueued so far on the current stream of the model's device."""
        with torch.cuda.device(self.device):
            ev = self.events[slot] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.events[slot] = ev

    def to_device(self, pilots: Optional[torch.Tensor], conds: Optional[List[torch.Tensor]]):
        """The training path's single asynchronous copy; returns device views (pilots | None, conds | None)."""
        slot, total, pil_h, cond_h = self.fill(pilots, conds)
        with torch.cuda.device(self.device):         # the copy belongs to the model's device, current or not
            dev = torch.empty(total, dtype=torch.uint8, device=self.device)
            dev.copy_(self.host[slot][:total], non_blocking=True)
        self.release(slot)
        base = self.host[slot].data_ptr()
        pil_dev = None
        if pil_h is not None:
            pil_dev = dev[:pil_h.numel() * 8].view(torch.complex64).view(pil_h.shape)
        if cond_h is None:
            return pil_dev, None
        offs = [c.data_ptr() - base for c in cond_h]
        return pil_dev, [dev[o: o + c.numel() * 4].view(torch.float32) for o, c in zip(offs, cond_h)]


class BaseFortiTranEstimator(nn.Module):
    """Pilot grid -> full OFDM grid estimator (reference src/models/fortitran.py:10-250)."""

    def __init__(self, system_config: SystemConfig, model_config: ModelConfig,
                 use_channel_adaptation: bool = False) -> None:
        super().__init__()
        self.system_config = system_config
        self.model_config = model_config
        self.use_channel_adaptation = use_channel_adaptation
        self.device = torch.device(model_config.device)
        self.logger = logging.getLogger(self.__class__.__name__)
        self._engine = None
        self._hip_precision = os.environ.get("AFT_PRECISION", "f32")   # see the hip_precision property
        self._engine_entries = ()      # (owner dict, key, tensor, data_ptr) per state_dict tensor of the engine
        self._stager = None            # pinned-ring H2D staging of CPU inputs on the HIP inference path
        self._hip_covered = False
        self._setup_dimensions()
        self._build_architecture()
        self.to(self.device)
        self._check_hip_coverage()
        self._log_initialization_info()

    # ---- construction (fortitran.py:52-126) -------------------------------------------------
    def _setup_dimensions(self) -> None:
        sc, mc = self.system_config, self.model_config
        self.ofdm_size = (sc.ofdm.num_scs, sc.ofdm.num_symbols)
        self.pilot_size = (sc.pilot.num_scs, sc.pilot.num_symbols)
        self.pilot_features = self.pilot_size[0] * self.pilot_size[1]
        self.ofdm_features = self.ofdm_size[0] * self.ofdm_size[1]
        self.patch_length = mc.patch_size[0] * mc.patch_size[1]
        self.transformer_input_dim = self.patch_length
        if self.use_channel_adaptation:
            if mc.adaptive_token_length is None:
                raise ValueError("adaptive_token_length must be set when channel adaptation is enabled")
            if mc.channel_adaptivity_hidden_sizes is None:
                raise ValueError("channel_adaptivity_hidden_sizes must be set when channel adaptation is enabled")
            if len(mc.channel_adaptivity_hidden_sizes) != 3:
                raise ValueError("channel_adaptivity_hidden_sizes must have exactly 3 values")
            self.transformer_input_dim += mc.adaptive_token_length
        check_shape_coupling(sc, mc, self.use_channel_adaptation)

    def _build_architecture(self) -> None:
        mc = self.model_config
        self.pilot_upsampler = HipLinear(self.pilot_features, self.ofdm_features)
        self.initial_enhancer = ConvEnhancer()
        self.patch_embedder = PatchEmbedding(tuple(mc.patch_size))
        if self.use_channel_adaptation:
            self.channel_adapter = ChannelAdapter(tuple(mc.channel_adaptivity_hidden_sizes))
        self.transformer_encoder = TransformerEncoderForChannels(
            input_dim=self.transformer_input_dim, output_dim=self.patch_length, model_dim=mc.model_dim,
            num_head=mc.num_head, activation=mc.activation, dropout=mc.dropout, num_layers=mc.num_layers,
            max_len=mc.max_seq_len, pos_encoding_type=mc.pos_encoding_type)
        self.patch_reconstructor = InversePatchEmbedding(self.ofdm_size, tuple(mc.patch_size))
        self.final_refiner = ConvEnhancer()

    def _log_initialization_info(self) -> None:
        info = self.get_model_info()
        self.logger.info("%s initialized: adaptation=%s ofdm=%s pilots=%s patch=%s d=%s layers=%s device=%s "
                         "params=%s trainable=%s", info["model_name"], info["channel_adaptation"], info["ofdm_size"],
                         info["pilot_size"], info["patch_size"], info["model_dim"], info["num_layers"],
                         info["device"], f"{info['total_parameters']:,}", f"{info['trainable_parameters']:,}")

    # ---- coverage + engine management ---------------------------------------------------------
    def _check_hip_coverage(self) -> None:
        """The single coverage predicate (module docstring): on a HIP device ask the library whether its
        kernels cover this configuration; refuse at construction if not."""
        self._hip_covered = False
        if self.pilot_upsampler.weight.device.type != "cuda":
            return
        from .hip_ops import config_coverage   # loads the extension: a missing .so raises here, loudly
        cfg = _abi.config_from_pydantic(self.system_config, self.model_config, self.use_channel_adaptation)
        reason = config_coverage(cfg)
        if reason is None:
            self._hip_covered = True
            # what loss.backward() runs on (no silent per-block fallback): logged once, also in training_backends()
            for block, gap in self.training_backends().items():
                if gap is not None:
                    self.logger.warning("training: %s is differentiated by PyTorch-ROCm autograd, not by the library's kernels (%s); "
                                        "inference (eval() + no_grad) runs the HIP engine either way", block, gap)
            return
        from . import _lib
        if _lib.get_switch("AFT_ALLOW_COMPOSITE") == "1":      # the AFT_* environment as the library read it at load, or aft_set_switch
            self.logger.warning("configuration not covered by the gfx950 kernels (%s): AFT_ALLOW_COMPOSITE=1, "
                                "running the PyTorch-ROCm composite for training AND evaluation", reason)
            return
        raise ValueError(f"configuration not covered by the gfx950 kernels: {reason}. Build the model with "
                         "device='cpu', or set AFT_ALLOW_COMPOSITE=1 to run the PyTorch-ROCm composite instead.")

    def hip_engine_name(self) -> Optional[str]:
        """``"packed"`` / ``"general"``: the launch sequence ``aft_forward_f32`` runs for this configuration (include/adafortitran_amd.h
        ``aft_engine_of``: a property of the configuration alone); None off the HIP device or when the configuration is not covered."""
        if not self._hip_covered:
            return None
        from . import _lib
        import ctypes
        cfg = _abi.config_from_pydantic(self.system_config, self.model_config, self.use_channel_adaptation)
        return {_abi.AFT_ENGINE_PACKED: "packed", _abi.AFT_ENGINE_GENERAL: "general"}.get(_lib.load().aft_engine_of(ctypes.byref(cfg)))

    def training_backends(self) -> dict:
        """{block: None | reason}: None = ``loss.backward()`` through that block runs this library's hand-written kernels on a HIP
        device; a string = it is differentiated by PyTorch-ROCm autograd and why (reference trainer.py:195-233 trains
        whatever encoders.py:44-51 builds).  The dense layers, the adapter MLPs and the loss glue are not listed: always HIP / torch."""
        from .hip_ops import conv_enhancer_covered
        conv_gap = None if conv_enhancer_covered(*self.ofdm_size) else f"no LDS band plan of the fused conv-stack kernel for a {self.ofdm_size} grid"
        return {"transformer_encoder": self.transformer_encoder.hip_train_gap(),
                "initial_enhancer / final_refiner": conv_gap}

    def _apply(self, fn, *args, **kwargs):  # .to()/.cuda()/.float() re-allocate parameters
        self._engine = None
        out = super()._apply(fn, *args, **kwargs)
        if hasattr(self, "pilot_upsampler") and self._hip_covered != (self.pilot_upsampler.weight.device.type == "cuda"):
            self._check_hip_coverage()    # moved between CPU and the HIP device after construction
        return out

    def load_state_dict(self, *args, **kwargs):
        self._engine = None
        return super().load_state_dict(*args, **kwargs)

    @property
    def hip_precision(self) -> str:
        """Arithmetic of the HIP inference path: ``"f32"`` (default: exact-fp32 MFMAs everywhere, the parity contract) or
        ``"bf16x3"`` -- the opt-in split-precision tier of include/adafortitran_amd.h (AFT_PRECISION_BF16X3: the encoder's
        GEMMs and attention products on bf16 hi/lo terms with fp32 accumulation; model_dim 128 or 256; ~1.8-2x the frames/s at
        max|d| ~ 3e-5 |y|max).  Not part of the reference's YAML surface: set it on the module, or AFT_PRECISION=bf16x3."""
        return self._hip_precision

    @hip_precision.setter
    def hip_precision(self, value: str) -> None:
        if value not in ("f32", "bf16x3"):
            raise ValueError("hip_precision must be 'f32' or 'bf16x3'")
        self._hip_precision = value
        self._engine = None

    def _hip_eligible(self) -> bool:
        return self._hip_covered and not self.training and not torch.is_grad_enabled()

    def _hip_engine(self):
        """The cached engine, or None when the parameters are not float32 (then the composite runs, as
        for any dtype the reference would run).  The per-call check is O(number of tensors) pointer /
        identity compares (~10 us): an engine stays valid until a tensor object is replaced in its
        module or re-homed (``p.data = ...``, e.g. by optim.FlatParameters); in-place updates by an
        optimizer keep it valid because the ABI reads the parameters' own storage."""
        eng = self._engine
        if eng is not None:
            for owner, key, tensor, ptr in self._engine_entries:
                cur = owner.get(key)
                if cur is not tensor or cur.data_ptr() != ptr:
                    eng = None
                    break
            if eng is not None:
                return eng
        from .hip_ops import HipEngine, config_coverage  # raises loudly if the extension is missing
        entries, tensors = [], {}
        for prefix, mod in self.named_modules():
            for owner in (mod._parameters, mod._buffers):
                for key, t in owner.items():
                    if t is None or (owner is mod._buffers and key in mod._non_persistent_buffers_set):
                        continue
                    tensors[f"{prefix}.{key}" if prefix else key] = t
                    entries.append((owner, key, t, t.data_ptr()))
        if any(t.is_floating_point() and t.dtype != torch.float32 for t in tensors.values()):
            self._engine, self._engine_entries = None, ()
            return None
        for k, v in tensors.items():
            if not v.is_contiguous():
                raise ValueError(f"parameter {k} is not contiguous")
        cfg = _abi.config_from_pydantic(self.system_config, self.model_config, self.use_channel_adaptation)
        if self._hip_precision == "bf16x3":
            cfg.precision = _abi.AFT_PRECISION_BF16X3
            reason = config_coverage(cfg)
            if reason is not None:
                raise ValueError(f"hip_precision='bf16x3': {reason}")
        eng = HipEngine(cfg, {k: v.detach() for k, v in tensors.items()})
        self._engine, self._engine_entries = eng, tuple(entries)
        return eng

    # ---- forward (fortitran.py:145-233) -----------------------------------------------------
    def forward(self, pilot_symbols: torch.Tensor, meta_data: Optional[Tuple] = None) -> torch.Tensor:
        if self.use_channel_adaptation and meta_data is None:
            raise ValueError("meta_data is required when channel adaptation is enabled")
        if not self.use_channel_adaptation and meta_data is not None:
            self.logger.warning("meta_data provided but channel adaptation is disabled - ignoring meta_data")
        conditions: Optional[List[torch.Tensor]] = None
        if self.use_channel_adaptation:
            _, snr, delay_spread, max_dop_shift, _, _ = meta_data
            conditions = [snr, delay_spread, max_dop_shift]

        eng = self._hip_engine() if self._hip_eligible() else None
        if eng is not None:
            return self._forward_hip(eng, pilot_symbols, conditions)
        # the model owns the H2D copy (fortitran.py:167-173): on a HIP device, CPU inputs as the DataLoader yields them go
        # through the pinned staging ring in one asynchronous copy (a pageable .to(device) blocks the host until the device
        # has drained the previous step, and the next forward's first launches then trickle in with the device idle --
        # 0.4-0.7 ms per training step); device-resident inputs pass through
        pilot_symbols, conditions = self._inputs_to_device(pilot_symbols, conditions)

        if pilot_symbols.device.type == "cuda" and torch.is_grad_enabled():
            # training on the HIP device: the Re and Im planes go through the network as ONE batch of
            # 2B planes (same per-sample arithmetic as the reference's two passes, fortitran.py:176-177),
            # which is what the encoder's training kernels are laid out for
            B = pilot_symbols.shape[0]
            stacked = torch.cat((pilot_symbols.real, pilot_symbols.imag), dim=0)
            cond2 = None if conditions is None else [torch.cat((c, c), dim=0) for c in conditions]
            out = self._forward_real_valued(stacked, cond2)
            # [2B,S,T] -> complex [B,S,T] through ONE copy (re / im interleaved, then a view): torch.complex(out[:B], out[B:]) costs
            # its backward two slice gradients (a zero fill + a copy each) and an add -- five small launches per step more
            return torch.view_as_complex(out.view(2, B, *out.shape[1:]).permute(1, 2, 3, 0).contiguous())
        real = self._forward_real_valued(pilot_symbols.real, conditions)
        imag = self._forward_real_valued(pilot_symbols.imag, conditions)
        return torch.complex(real, imag)

    def _stageable(self, pilot_symbols: torch.Tensor, conditions: Optional[List[torch.Tensor]]):
        """Which inputs can go through the pinned ring: CPU complex64 [B, ., .] pilots, CPU float32 conditions of B values."""
        B = pilot_symbols.shape[0] if pilot_symbols.dim() == 3 else -1
        pil_cpu = pilot_symbols.device.type == "cpu" and pilot_symbols.dtype == torch.complex64 and B > 0
        cond_cpu = conditions is not None and B > 0 and all(
            c.device.type == "cpu" and c.dtype == torch.float32 and c.numel() == B for c in conditions)
        return pil_cpu, cond_cpu

    def _ring(self, dev: torch.device) -> _InputStager:
        if self._stager is None or self._stager.device != dev:
            self._stager = _InputStager(dev)
        return self._stager

    def _forward_hip(self, eng, pilot_symbols: torch.Tensor, conditions: Optional[List[torch.Tensor]]) -> torch.Tensor:
        """The inference path: ONE stateless ``aft_forward_f32`` call (the encoder weights are re-laid into fragment order
        inside it, in the same launch as the channel adapter -- nothing cached, nothing that could go stale).  CPU inputs
        are placed in a pinned ring slot that the kernels read directly; no copy is enqueued (see _InputStager)."""
        pil_cpu, cond_cpu = self._stageable(pilot_symbols, conditions)
        if not (pil_cpu or cond_cpu):
            if conditions is not None:
                conditions = [t.to(self.device) for t in conditions]
            return eng.forward(pilot_symbols.to(self.device), *(conditions or ()))
        ring = self._ring(eng.device)
        slot, _, pil_h, cond_h = ring.fill(pilot_symbols if pil_cpu else None, conditions if cond_cpu else None)
        try:
            pil = pil_h if pil_cpu else pilot_symbols.to(self.device)
            conds = cond_h if cond_cpu else (None if conditions is None else [t.to(self.device) for t in conditions])
            return eng.forward(pil, *(conds or ()), pinned_inputs=True)
        finally:
            ring.release(slot)

    def _inputs_to_device(self, pilot_symbols: torch.Tensor, conditions: Optional[List[torch.Tensor]]):
        dev = self.pilot_upsampler.weight.device
        pil_cpu, cond_cpu = self._stageable(pilot_symbols, conditions)
        if dev.type == "cuda" and (pil_cpu or cond_cpu):
            pil, conds = self._ring(dev).to_device(pilot_symbols if pil_cpu else None, conditions if cond_cpu else None)
            if pil is None:
                pil = pilot_symbols.to(self.device)
            if conds is not None:   # keep each condition's [B, 1] / [B] shape
                conds = [c.view(o.shape) for c, o in zip(conds, conditions)]
            elif conditions is not None:
                conds = [t.to(self.device) for t in conditions]
            return pil, conds
        if conditions is not None:
            conditions = [t.to(self.device) for t in conditions]
        return pilot_symbols.to(self.device), conditions

    def _forward_real_valued(self, x: torch.Tensor, channel_conditions: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        """Differentiable composite of stages S1-S8 on one real plane batch [B,Ps,Pt]."""
        B = x.shape[0]
        grid = self.pilot_upsampler(x.reshape(B, -1)).view(B, 1, *self.ofdm_size)
        conv_enhanced = self.initial_enhancer(grid).squeeze(1)
        adaptive = self.use_channel_adaptation and channel_conditions is not None
        enc, patch = self.transformer_encoder, self.patch_embedder.patch_size
        if enc.fused_ends_ok(conv_enhanced, patch) and enc.linear_1.in_features == patch[0] * patch[1] + (6 if adaptive else 0):
            # HIP training path: embedding and reconstruction as one launch each (training.HipEmbedFunction / HipTailFunction)
            combined = enc.forward_planes(conv_enhanced, self.channel_adapter(*channel_conditions) if adaptive else None, patch)
        else:
            tokens = self.patch_embedder(conv_enhanced)
            if adaptive:
                tokens = torch.cat((tokens, self.channel_adapter(*channel_conditions)), dim=2)
            encoded = self.transformer_encoder(tokens)
            combined = conv_enhanced + self.patch_reconstructor(encoded)
        return self.final_refiner(combined.unsqueeze(1)).squeeze(1)

    def get_model_info(self) -> dict:
        mc = self.model_config
        return {
            "model_name": self.__class__.__name__,
            "channel_adaptation": self.use_channel_adaptation,
            "ofdm_size": self.ofdm_size,
            "pilot_size": self.pilot_size,
            "patch_size": mc.patch_size,
            "patch_length": self.patch_length,
            "transformer_input_dim": self.transformer_input_dim,
            "model_dim": mc.model_dim,
            "num_layers": mc.num_layers,
            "device": str(self.device),
            "total_parameters": sum(p.numel() for p in self.parameters()),
            "trainable_parameters": sum(p.numel() for p in self.parameters() if p.requires_grad),
        }


class FortiTranEstimator(BaseFortiTranEstimator):
    """No channel adaptation (reference src/models/fortitran.py:253-268)."""

    def __init__(self, system_config: SystemConfig, model_config: ModelConfig) -> None:
        super().__init__(system_config, model_config, use_channel_adaptation=False)


class AdaFortiTranEstimator(BaseFortiTranEstimator):
    """Adaptive tokens from SNR / delay spread / Doppler (reference src/models/adafortitran.py:5-22)."""

    def __init__(self, system_config: SystemConfig, model_config: ModelConfig) -> None:
        super().__init__(system_config, model_config, use_channel_adaptation=True)


class LinearEstimator(nn.Module):
    """Learned linear map pilots -> grid (reference src/models/linear.py:15-106).

    The reference's real ``nn.Linear`` raises on the complex64 input its own dataset
    produces.  This estimator accepts what the reference accepts (real
    [B,Ps,Pt]) and additionally complex64, applying the real map to Re and Im separately
    -- the only reading consistent with fortitran.py:176-180."""

    def __init__(self, system_config: SystemConfig, model_config: ModelConfig) -> None:
        super().__init__()
        self.system_config = system_config
        self.model_config = model_config
        self.device = torch.device(model_config.device)
        self.logger = logging.getLogger(__name__)
        self.ofdm_size = (system_config.ofdm.num_scs, system_config.ofdm.num_symbols)
        self.pilot_size = (system_config.pilot.num_scs, system_config.pilot.num_symbols)
        self.linear = nn.Linear(self.pilot_size[0] * self.pilot_size[1], self.ofdm_size[0] * self.ofdm_size[1])
        self.to(self.device)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        expected = (x.size(0), self.pilot_size[0], self.pilot_size[1])
        if tuple(x.size()) != expected:
            raise ValueError(f"Expected input shape {expected}, got {x.size()}")
        x = x.to(self.device)
        if x.is_complex():
            w = self.linear.weight
            if (w.device.type == "cuda" and not torch.is_grad_enabled() and x.dtype == torch.complex64
                    and w.dtype == torch.float32):
                from .hip_ops import linear_forward
                return linear_forward(w.detach(), self.linear.bias.detach(), x, self.ofdm_size)
            return torch.complex(self._plane(x.real), self._plane(x.imag))
        return self._plane(x)

    def _plane(self, x: torch.Tensor) -> torch.Tensor:
        return self.linear(torch.flatten(x, start_dim=1)).reshape(-1, *self.ofdm_size)

    def __repr__(self) -> str:
        return f"LinearEstimator(\n  ofdm_size={self.ofdm_size},\n  pilot_size={self.pilot_size},\n  device={self.device}\n)"
