"""Deterministic synthetic weights and inputs (no dataset / checkpoint ships
with the reference).

Everything here is integer-hash based (splitmix64 on ``(stream, index)``), so
the same ``seed`` gives bit-identical float32 tensors on every machine: the
fixture generator (run where the reference can be imported) and the GPU box
(where it cannot) regenerate the *same* weights instead of shipping 4 MB files.

Tensor names and shapes are the reference's ``state_dict`` layout
(reference src/models/fortitran.py:83-126,
blocks/encoders.py:36-56, blocks/channel_adaptivity.py:35-39,
blocks/enhancers.py:12-20).  Scales follow PyTorch's default initialisers
(uniform +-1/sqrt(fan_in) for Linear/Conv, Xavier for the packed in-proj,
std 0.02 for the positional table) with optional gains to build "hot"
variants whose softmax is far from uniform.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def _stream_id(seed: int, name: str) -> np.uint64:
    return np.uint64(((seed & 0xFFFFFFFF) << 32) | (zlib.crc32(name.encode()) & 0xFFFFFFFF))


def hash_uniform(seed: int, name: str, n: int, lane: int = 0) -> np.ndarray:
    """``n`` float64 values, each an exact multiple of 2^-24 in [0, 1)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([_stream_id(seed, name)], dtype=np.uint64)
                           + np.uint64(lane) * np.uint64(0xD1B54A32D192ED03))[0]
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64((idx * np.uint64(0x2545F4914F6CDD1D) + base) & _M64)
    return (bits >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


def uniform_pm(seed: int, name: str, shape: Tuple[int, ...], bound: float) -> np.ndarray:
    """float32 tensor uniform in [-bound, bound)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(seed, name, n)
    return ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape)


def normal_like(seed: int, name: str, shape: Tuple[int, ...], std: float = 1.0) -> np.ndarray:
    """Approximately N(0, std^2): Irwin-Hall sum of 12 hash uniforms (exact in
    float64, hence bit-reproducible), cast to float32."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float64)
    for lane in range(12):
        acc += hash_uniform(seed, name, n, lane=lane + 1)
    return ((acc - 6.0) * std).astype(np.float32).reshape(shape)


def token_count(num_scs: int, num_symbols: int, patch: Tuple[int, int]) -> int:
    return (num_scs // patch[0]) * (num_symbols // patch[1])


def make_state_dict(*, ofdm: Tuple[int, int], pilot: Tuple[int, int], patch: Tuple[int, int],
                    num_layers: int, model_dim: int, num_head: int, max_seq_len: int = 512,
                    adaptive_hidden: Optional[Tuple[int, int, int]] = None,
                    pos_encoding_type: str = "learnable",
                    seed: int = 20251114, attn_gain: float = 1.0, ffn_gain: float = 1.0,
                    head_gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """All tensors of the (Ada)FortiTran ``state_dict`` as float32 numpy arrays.

    ``adaptive_hidden`` given -> AdaFortiTran (adds ``channel_adapter.*`` and a
    12-wide ``linear_1``); ``None`` -> FortiTran.  ``attn_gain`` scales the packed
    in-proj weights (logits grow with its square), ``ffn_gain`` the FFN weights,
    ``head_gain`` the final ``linear_2`` so the encoder output is not negligible
    next to the residual.
    """
    S, T = ofdm
    Ps, Pt = pilot
    p = patch[0] * patch[1]
    d = model_dim
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def lin(prefix: str, n_out: int, n_in: int, gain: float = 1.0):
        bound = 1.0 / np.sqrt(n_in)
        sd[prefix + ".weight"] = uniform_pm(seed, prefix + ".weight", (n_out, n_in), bound * gain)
        sd[prefix + ".bias"] = uniform_pm(seed, prefix + ".bias", (n_out,), bound)

    def conv_stack(prefix: str):
        for slot, (cin, cout) in zip((0, 2, 4, 6), ((1, 8), (8, 32), (32, 8), (8, 1))):
            bound = 1.0 / np.sqrt(cin * 9)
            name = f"{prefix}.conv_block.{slot}"
            sd[name + ".weight"] = uniform_pm(seed, name + ".weight", (cout, cin, 3, 3), bound)
            sd[name + ".bias"] = uniform_pm(seed, name + ".bias", (cout,), bound)

    lin("pilot_upsampler", S * T, Ps * Pt)
    conv_stack("initial_enhancer")
    if adaptive_hidden is not None:
        h0, h1, h2 = adaptive_hidden
        for enc in ("snr_encoder", "ds_encoder", "dop_encoder"):
            for slot, (n_in, n_out) in zip((0, 2, 4), ((1, h0), (h0, h1), (h1, h2))):
                lin(f"channel_adapter.{enc}.{slot}", n_out, n_in)
    te = "transformer_encoder"
    lin(te + ".linear_1", d, p + (6 if adaptive_hidden is not None else 0))
    if pos_encoding_type == "learnable":
        sd[te + ".positional_encoding.position_embeddings"] = uniform_pm(
            seed, te + ".pos", (1, max_seq_len, d), 0.02 * np.sqrt(3.0))
    else:
        sd[te + ".positional_encoding.pe"] = sinusoid_table(max_seq_len, d)
    for i in range(num_layers):
        lp = f"{te}.transformer.layers.{i}"
        xav = np.sqrt(6.0 / (d + 3 * d))
        sd[lp + ".self_attn.in_proj_weight"] = uniform_pm(seed, lp + ".inw", (3 * d, d), xav * attn_gain)
        sd[lp + ".self_attn.in_proj_bias"] = uniform_pm(seed, lp + ".inb", (3 * d,), 0.05)
        lin(lp + ".self_attn.out_proj", d, d)
        lin(lp + ".linear1", 2 * d, d, ffn_gain)
        lin(lp + ".linear2", d, 2 * d, ffn_gain)
        for nm in ("norm1", "norm2"):
            sd[f"{lp}.{nm}.weight"] = (1.0 + uniform_pm(seed, f"{lp}.{nm}.w", (d,), 0.1)).astype(np.float32)
            sd[f"{lp}.{nm}.bias"] = uniform_pm(seed, f"{lp}.{nm}.b", (d,), 0.05)
    lin(te + ".linear_2", p, d, head_gain)
    conv_stack("final_refiner")
    return sd


def sinusoid_table(max_len: int, d_model: int) -> np.ndarray:
    """Fixed sin/cos table, sin on even columns, cos on odd, base 10000
    (reference blocks/positional_encodings.py:15-24), evaluated in float32 the
    way the reference evaluates it (float32 exp / sin / cos via torch)."""
    import torch
    position = torch.arange(0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-torch.log(torch.tensor(10000.0)) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe.numpy()


_SNR_GRID = np.arange(0, 31, 5, dtype=np.float32)          # data/test/SNR_test_set/SNR_*
_DS_GRID = np.arange(50, 351, 50, dtype=np.float32)        # data/test/DS_test_set/DS_*
_DOP_GRID = np.arange(200, 1401, 200, dtype=np.float32)    # data/test/MDS_test_set/DOP_*


def make_inputs(batch: int, *, ofdm: Tuple[int, int] = (120, 14), pilot: Tuple[int, int] = (12, 2),
                seed: int = 20251114) -> Dict[str, np.ndarray]:
    """Synthetic batch in the format the reference's dataset hands the model
    (reference src/data/dataset.py:125-139, src/utils.py:102-108):
    ``pilots`` complex64 [B,Ps,Pt], ``target`` complex64 [B,S,T] and raw,
    un-normalised ``snr``/``ds``/``dop`` float32 [B,1] drawn from the test-folder grids."""
    Ps, Pt = pilot
    S, T = ofdm
    pr = normal_like(seed, "pilots.re", (batch, Ps, Pt))
    pi = normal_like(seed, "pilots.im", (batch, Ps, Pt))
    tr = normal_like(seed, "target.re", (batch, S, T))
    ti = normal_like(seed, "target.im", (batch, S, T))

    def pick(name: str, grid: np.ndarray) -> np.ndarray:
        u = hash_uniform(seed, name, batch)
        return grid[np.minimum((u * len(grid)).astype(np.int64), len(grid) - 1)].reshape(batch, 1)

    return {
        "pilots": (pr + 1j * pi).astype(np.complex64),
        "target": (tr + 1j * ti).astype(np.complex64),
        "snr": pick("meta.snr", _SNR_GRID),
        "ds": pick("meta.ds", _DS_GRID),
        "dop": pick("meta.dop", _DOP_GRID),
    }


def meta_tuple(inputs: Dict[str, np.ndarray]):
    """The default-collated 6-tuple the reference's DataLoader yields
    (file_no, snr, ds, dop, n, names); only indices 1..3 are read
    (reference src/models/fortitran.py:166)."""
    import torch
    b = inputs["snr"].shape[0]
    z = torch.zeros(b, 1)
    return (z, torch.from_numpy(inputs["snr"]), torch.from_numpy(inputs["ds"]),
            torch.from_numpy(inputs["dop"]), z, [tuple(str(i) for i in range(b))])


def state_dict_checksum(sd: Dict[str, np.ndarray]) -> str:
    """CRC over the raw bytes of every tensor, in key order (fixture metadata)."""
    crc = 0
    for k, v in sd.items():
        if k.endswith(".pe"):  # sin/cos tables come from libm: not bit-reproducible across hosts
            continue
        crc = zlib.crc32(k.encode(), crc)
        crc = zlib.crc32(np.ascontiguousarray(v).tobytes(), crc)
    return f"{crc:08x}"
