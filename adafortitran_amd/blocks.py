"""Parameter containers + the differentiable (training) forward of the estimator's building blocks.

These nn.Modules exist for two reasons: (1) they own the parameters under exactly the
reference's ``state_dict`` names, so checkpoints move both ways;
(2) their ``forward`` is the differentiable path used under autograd: on the HIP device the encoder
layers, the conv stacks, the dense layers and the adapter MLPs call the library's hand-written
forward/backward kernels through ``torch.autograd.Function`` (training.py); on CPU
they are the plain ``torch.nn`` modules the reference is made of.
In ``eval()`` + ``no_grad()`` on a HIP device the estimator bypasses these forwards and
calls the C ABI (``aft_forward_f32``) on the parameters' device pointers.

Behavioural spec: reference src/models/blocks/{enhancers,patch_processors,
channel_adaptivity,encoders,positional_encodings}.py.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch
from torch import nn
import torch.nn.functional as F

from .training import HipLinear   # nn.Linear whose grad-enabled forward on the HIP device runs the library's GEMMs


class ConvEnhancer(nn.Module):
    """3x3 conv stack 1->8->32->8->1 (ReLU between) under the key ``conv_block.{0,2,4,6}``
    (reference blocks/enhancers.py:12-20)."""

    WIDTHS = (1, 8, 32, 8, 1)

    def __init__(self) -> None:
        super().__init__()
        layers = []
        for i, (cin, cout) in enumerate(zip(self.WIDTHS[:-1], self.WIDTHS[1:])):
            layers.append(nn.Conv2d(cin, cout, kernel_size=3, padding=1))
            if i < len(self.WIDTHS) - 2:
                layers.append(nn.ReLU())
        self.conv_block = nn.Sequential(*layers)

    #: set to False to differentiate the stack through PyTorch-ROCm (MIOpen) instead (A/B tests)
    hip_training = True
    _covered = {}   # (S, T) -> the fused kernel has an LDS band plan for that grid

    @classmethod
    def _grid_covered(cls, S: int, T: int) -> bool:
        key = (int(S), int(T))
        if key not in cls._covered:
            from .hip_ops import conv_enhancer_covered
            cls._covered[key] = conv_enhancer_covered(*key)
        return cls._covered[key]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (self.hip_training and x.device.type == "cuda" and torch.is_grad_enabled() and x.dtype == torch.float32
                and x.dim() == 4 and x.shape[1] == 1 and self._grid_covered(x.shape[2], x.shape[3])):
            # grad-enabled forward on the HIP device: the fused conv-stack kernel, its own backward
            from .training import HipConvEnhancerFunction
            convs = [self.conv_block[i] for i in (0, 2, 4, 6)]
            args = [t for c in convs for t in (c.weight, c.bias)]
            return HipConvEnhancerFunction.apply(x, *args)
        return self.conv_block(x)


class PatchEmbedding(nn.Module):
    """[B,S,T] -> [B,tokens,p0*p1]; token = (sc//p0)*(T//p1) + sym//p1, feature =
    (sc%p0)*p1 + sym%p1 (what nn.Unfold(kernel=stride=patch)+permute yields,
    reference blocks/patch_processors.py:22,34-35) -- done here as a pure view/permute."""

    def __init__(self, patch_size: Tuple[int, int] = (3, 2)) -> None:
        super().__init__()
        self.patch_size = tuple(patch_size)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        p0, p1 = self.patch_size
        B, S, T = x.shape
        return x.reshape(B, S // p0, p0, T // p1, p1).permute(0, 1, 3, 2, 4).reshape(B, (S // p0) * (T // p1), p0 * p1)


class InversePatchEmbedding(nn.Module):
    """Exact inverse of :class:`PatchEmbedding` (reference patch_processors.py:53-57,69-71:
    non-overlapping Fold sums nothing)."""

    def __init__(self, output_size: Tuple[int, int] = (120, 14), patch_size: Tuple[int, int] = (3, 2)) -> None:
        super().__init__()
        self.output_size = tuple(output_size)
        self.patch_size = tuple(patch_size)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        (S, T), (p0, p1) = self.output_size, self.patch_size
        B = x.shape[0]
        return x.reshape(B, S // p0, T // p1, p0, p1).permute(0, 1, 3, 2, 4).reshape(B, S, T)


class ChannelAdapter(nn.Module):
    """Three MLPs (snr / ds / dop) 1->h0->h1->h2, ReLU between; each output viewed as
    [B, h2/2, 2] and concatenated on the last axis -> [B, tokens, 6]
    (reference blocks/channel_adaptivity.py:24-40,59-63)."""

    def __init__(self, hidden_sizes: Sequence[int]) -> None:
        super().__init__()
        h0, h1, h2 = hidden_sizes

        def mlp() -> nn.Sequential:
            return nn.Sequential(HipLinear(1, h0), nn.ReLU(), HipLinear(h0, h1), nn.ReLU(), HipLinear(h1, h2))

        self.snr_encoder, self.ds_encoder, self.dop_encoder = mlp(), mlp(), mlp()

    #: set to False to differentiate the three MLPs layer by layer (HipLinear / PyTorch-ROCm) instead
    hip_training = True

    def _hip_train_eligible(self, x: torch.Tensor) -> bool:
        first, second = self.snr_encoder[0], self.snr_encoder[2]
        return (self.hip_training and x.device.type == "cuda" and torch.is_grad_enabled()
                and first.weight.dtype == torch.float32 and first.out_features <= 256 and second.out_features <= 64)

    def forward(self, snr: torch.Tensor, delay_spread: torch.Tensor, doppler_shift: torch.Tensor) -> torch.Tensor:
        if self._hip_train_eligible(snr):
            # grad-enabled forward on the HIP device: one forward and two backward kernels for all three MLPs
            from .training import HipChannelAdapterFunction
            params = [t for enc in (self.snr_encoder, self.ds_encoder, self.dop_encoder) for i in (0, 2, 4)
                      for t in (enc[i].weight, enc[i].bias)]
            return HipChannelAdapterFunction.apply(snr, delay_spread, doppler_shift, *params)
        B = snr.shape[0]
        parts = [enc(v).reshape(B, -1, 2) for enc, v in ((self.snr_encoder, snr), (self.ds_encoder, delay_spread),
                                                        (self.dop_encoder, doppler_shift))]
        return torch.cat(parts, dim=2)


class SinusoidalPositionalEncoding(nn.Module):
    """Fixed table in buffer ``pe`` [1,max_len,d]: sin on even columns, cos on odd, base 1e4
    (reference blocks/positional_encodings.py:5-38)."""

    def __init__(self, max_len: int, d_model: int) -> None:
        super().__init__()
        pos = torch.arange(0, max_len).unsqueeze(1)
        freq = torch.exp(torch.arange(0, d_model, 2) * (-torch.log(torch.tensor(10000.0)) / d_model))
        table = torch.zeros(1, max_len, d_model)
        table[0, :, 0::2] = torch.sin(pos * freq)
        table[0, :, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", table)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x + self.pe[:, : x.size(1), :]


class LearnablePositionalEncoding(nn.Module):
    """Parameter ``position_embeddings`` [1,max_len,d], trunc-normal(std=0.02)
    (reference blocks/positional_encodings.py:41-64)."""

    def __init__(self, max_len: int, d_model: int) -> None:
        super().__init__()
        self.position_embeddings = nn.Parameter(torch.zeros(1, max_len, d_model))
        nn.init.trunc_normal_(self.position_embeddings, std=0.02)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x + self.position_embeddings[:, : x.size(1), :]


class TransformerEncoderForChannels(nn.Module):
    """linear_1 -> + positional table -> L x post-LN encoder layers (FFN = 2d) -> linear_2
    (reference blocks/encoders.py:7-70)."""

    def __init__(self, input_dim: int, output_dim: int, model_dim: int = 128, num_head: int = 4,
                 activation: str = "gelu", dropout: float = 0.1, num_layers: int = 3, max_len: int = 512,
                 pos_encoding_type: str = "learnable") -> None:
        super().__init__()
        self.linear_1 = HipLinear(input_dim, model_dim)
        if pos_encoding_type == "learnable":
            self.positional_encoding = LearnablePositionalEncoding(max_len, model_dim)
        elif pos_encoding_type == "sinusoidal":
            self.positional_encoding = SinusoidalPositionalEncoding(max_len, model_dim)
        else:
            raise ValueError("pos_encoding_type must be 'learnable' or 'sinusoidal'")
        layer = nn.TransformerEncoderLayer(d_model=model_dim, nhead=num_head, dim_feedforward=2 * model_dim,
                                           activation=activation, dropout=dropout, batch_first=True)
        self.transformer = nn.TransformerEncoder(layer, num_layers=num_layers)
        self.linear_2 = HipLinear(model_dim, output_dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = self.positional_encoding(self.linear_1(x))
        return self.linear_2(self.run_layers(h))

    #: the two thin ends (patch embedding + linear_1 + positions; linear_2 + inverse patch embedding + residual) as one launch each in
    #: the HIP training path (training.HipEmbedFunction / HipTailFunction); False = PyTorch's unfold / cat / add around HipLinear (A/B)
    fused_ends = True

    def positional_table(self) -> torch.Tensor:
        pe = self.positional_encoding
        return pe.position_embeddings if hasattr(pe, "position_embeddings") else pe.pe

    def fused_ends_ok(self, conv_enhanced: torch.Tensor, patch) -> bool:
        """True when the estimator may hand this encoder the conv-enhanced planes themselves (``forward_planes``)."""
        from . import _lib
        from .training import HipLinear
        p = patch[0] * patch[1]
        l1, l2, d = self.linear_1, self.linear_2, self.linear_1.out_features
        on = lambda lin: HipLinear.default_hip_training if lin.hip_training is None else lin.hip_training
        return (self.fused_ends and not _lib.get_switch("AFT_TRAIN_NO_FUSED_ENDS") and conv_enhanced.dim() == 3
                and conv_enhanced.device.type == "cuda" and torch.is_grad_enabled() and conv_enhanced.dtype == torch.float32
                and conv_enhanced.shape[0] % 2 == 0 and self.hip_train_gap() is None and on(l1) and on(l2)
                and l1.bias is not None and l2.bias is not None and l1.weight.dtype == torch.float32
                and p <= 32 and d % 4 == 0 and d <= 512 and l2.out_features == p and l1.in_features in (p, p + 6))

    def forward_planes(self, conv_enhanced: torch.Tensor, adapter_tokens: Optional[torch.Tensor], patch) -> torch.Tensor:
        """conv_enhanced [P,S,T] (+ adapter tokens [P,tokens,6]) -> conv_enhanced + InversePatchEmbedding(encoder(tokens)): what the
        estimator computes around this module (reference fortitran.py:212-227), with both thin ends on the library's kernels."""
        from .training import HipEmbedFunction, HipTailFunction
        h = HipEmbedFunction.apply(conv_enhanced, adapter_tokens, self.linear_1.weight, self.linear_1.bias, self.positional_table(),
                                   tuple(patch))
        h = self.run_layers(h)
        return HipTailFunction.apply(h, self.linear_2.weight, self.linear_2.bias, conv_enhanced, tuple(patch))

    def run_layers(self, h: torch.Tensor) -> torch.Tensor:
        if self._hip_train_eligible(h):
            # grad-enabled forward on the HIP device: hand-written forward/backward kernels for the
            # encoder layers (training.py); everything else differentiates through PyTorch-ROCm.
            from . import _abi
            from .training import encoder_stack_train
            layer0 = self.transformer.layers[0]
            cfg = _abi.make_config(ofdm=(h.shape[1], 1), pilot=(1, 1), patch=(1, 1), num_layers=len(self.transformer.layers),
                                   model_dim=h.shape[2], num_head=layer0.self_attn.num_heads,
                                   activation="gelu" if layer0.activation is F.gelu else "relu")
            return encoder_stack_train(h, list(self.transformer.layers), cfg, layer0.dropout.p if self.training else 0.0)
        return self.transformer(h)

    #: set to False to differentiate the encoder through PyTorch-ROCm autograd instead (A/B tests)
    hip_training = True

    def hip_train_gap(self) -> Optional[str]:
        """None when ``loss.backward()`` through this encoder runs the library's training kernels on a HIP device, else why it is
        differentiated by PyTorch-ROCm autograd instead (the estimator logs this once at construction: no silent fallback).
        Covered: every model_dim the inference engines take (multiples of 8 up to 512: the fused row-local kernels at 128, launch
        sequences elsewhere), every head dim up to 128 (32 is the kernels' own shape; 64 / 96 / 128 run two / three / four 32-feature
        blocks per head, anything else as zero-padded heads of the next multiple of 32), any token count, gelu / relu, post-norm,
        dim_feedforward = 2 model_dim (what the reference builds, encoders.py:44-51)."""
        layer0 = self.transformer.layers[0]
        d, heads = layer0.self_attn.embed_dim, layer0.self_attn.num_heads
        if not self.hip_training:
            return "hip_training is switched off on this module"
        if d % 8 or not 8 <= d <= 512:
            return f"model_dim {d} is not a multiple of 8 up to 512"
        if d % heads or d // heads > 128:
            return f"head dim {d / heads:g}: the attention kernels cover heads of up to 128 features"
        if layer0.activation not in (F.gelu, F.relu) or layer0.norm_first or layer0.linear1.out_features != 2 * d:
            return "encoder layer is not the reference's post-norm gelu / relu layer with dim_feedforward = 2 model_dim"
        return None

    def _hip_train_eligible(self, h: torch.Tensor) -> bool:
        if not (h.device.type == "cuda" and torch.is_grad_enabled() and h.dtype == torch.float32):
            return False
        return self.hip_train_gap() is None and h.shape[0] % 2 == 0


__all__ = ["ChannelAdapter", "TransformerEncoderForChannels", "ConvEnhancer", "PatchEmbedding",
           "InversePatchEmbedding", "SinusoidalPositionalEncoding", "LearnablePositionalEncoding"]
