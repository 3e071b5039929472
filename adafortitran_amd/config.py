"""Configuration surface of the estimators (YAML -> validated objects).

Mirrors the *field names, defaults and error behaviour* of the reference's
pydantic models so that the reference's own ``config/*.yaml`` files and its
trainer (which reads ``model_config.device``, ``.patch_size`` ...) keep working:

* ``SystemConfig`` / ``OFDMParams`` / ``PilotParams``  <- reference src/config/schemas.py:6-45
* ``ModelConfig`` (+ device validation)               <- reference src/config/schemas.py:48-175
* ``load_config(system_yaml, model_yaml)``             <- reference src/config/config_loader.py:16-78

Unknown keys are rejected (``extra="forbid"``), pilots may not exceed the OFDM
grid, AdaFortiTran needs both adaptive fields and FortiTran must not carry them.
On top of the reference behaviour, :func:`check_shape_coupling` turns the shape
couplings the reference only trips over at the first forward
into a ``ValueError`` at construction time.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import List, Literal, Optional, Tuple, Union

import torch
import yaml
from pydantic import BaseModel, ConfigDict, Field, ValidationError, model_validator

_log = logging.getLogger(__name__)


class OFDMParams(BaseModel):
    num_scs: int = Field(..., gt=0, description="OFDM subcarriers")
    num_symbols: int = Field(..., gt=0, description="OFDM symbols")


class PilotParams(BaseModel):
    num_scs: int = Field(..., gt=0, description="pilot subcarriers")
    num_symbols: int = Field(..., gt=0, description="pilot symbols")


class SystemConfig(BaseModel):
    """OFDM grid + pilot grid; pilots must fit inside the grid."""

    model_config = ConfigDict(extra="forbid")

    ofdm: OFDMParams
    pilot: PilotParams

    @model_validator(mode="after")
    def _pilots_fit(self):
        for axis, what in (("num_scs", "sub-carriers"), ("num_symbols", "symbols")):
            p, o = getattr(self.pilot, axis), getattr(self.ofdm, axis)
            if p > o:
                raise ValueError(f"Pilot {what} ({p}) cannot exceed OFDM {what} ({o})")
        return self


def _known_devices() -> List[str]:
    names = ["cpu"]
    if torch.cuda.is_available():
        names += ["cuda"] + [f"cuda:{i}" for i in range(torch.cuda.device_count())]
    if getattr(torch.backends, "mps", None) is not None and torch.backends.mps.is_available():
        names.append("mps")
    return names


def resolve_device(spec: str) -> str:
    """Validate a device string the way the reference's BaseConfig does
    (schemas.py:53-110).  ``cuda`` is the MI355X spelling under PyTorch-ROCm."""
    low = spec.lower()
    if low == "auto":
        known = _known_devices()
        return "cuda" if "cuda" in known else ("mps" if "mps" in known else "cpu")
    if low == "cpu":
        return spec
    if low.startswith("cuda"):
        if not torch.cuda.is_available():
            raise ValueError("CUDA is not available on this system")
        if ":" in low:
            tail = low.split(":", 1)[1]
            if not tail.isdigit():
                raise ValueError(f"Invalid CUDA device format: {low}")
            if int(tail) >= torch.cuda.device_count():
                raise ValueError(
                    f"CUDA device {int(tail)} not available. "
                    f"Available CUDA devices: {list(range(torch.cuda.device_count()))}")
        return spec
    if low == "mps":
        if "mps" not in _known_devices():
            raise ValueError("MPS is not available/detected on this system")
        return spec
    raise ValueError(f"Unsupported device: '{spec}'. Available devices: {_known_devices()}")


class BaseConfig(BaseModel):
    device: str = Field(default="cpu", description="Computing device to use")

    @model_validator(mode="after")
    def _device_ok(self):
        self.device = resolve_device(self.device)
        return self


class ModelConfig(BaseConfig):
    """Architecture hyper-parameters (same keys as the reference YAMLs)."""

    model_config = ConfigDict(extra="forbid")

    model_type: Literal["linear", "fortitran", "adafortitran"] = "fortitran"
    patch_size: Tuple[int, int]
    num_layers: int = Field(..., gt=0)
    model_dim: int = Field(..., gt=0)
    num_head: int = Field(..., gt=0)
    activation: Literal["relu", "gelu"] = "gelu"
    dropout: float = Field(default=0.1, ge=0.0, le=1.0)
    max_seq_len: int = Field(default=512, gt=0)
    pos_encoding_type: Literal["learnable", "sinusoidal"] = "learnable"
    adaptive_token_length: Optional[int] = Field(default=None, gt=0)
    channel_adaptivity_hidden_sizes: Optional[List[int]] = None

    @model_validator(mode="after")
    def _per_model_fields(self):
        adaptive_fields = {
            "channel_adaptivity_hidden_sizes": self.channel_adaptivity_hidden_sizes,
            "adaptive_token_length": self.adaptive_token_length,
        }
        if self.model_type == "adafortitran":
            for name, val in adaptive_fields.items():
                if val is None:
                    raise ValueError(f"{name} is required for AdaFortiTran model")
        elif self.model_type == "fortitran":
            for name, val in adaptive_fields.items():
                if val is not None:
                    raise ValueError(f"{name} should not be provided for FortiTran model")
        return self


def check_shape_coupling(system_config: SystemConfig, model_config: ModelConfig,
                         adaptive: bool) -> None:
    """Construction-time check of couplings the reference leaves to a late
    ``RuntimeError``: grid divisible by the patch, positional
    table long enough, adapter width == 2 x tokens and 6 adaptive features."""
    S, T = system_config.ofdm.num_scs, system_config.ofdm.num_symbols
    p0, p1 = model_config.patch_size
    if p0 <= 0 or p1 <= 0 or S % p0 or T % p1:
        raise ValueError(f"OFDM grid {S}x{T} is not divisible by patch_size {(p0, p1)}")
    tokens = (S // p0) * (T // p1)
    if model_config.max_seq_len < tokens:
        raise ValueError(f"max_seq_len ({model_config.max_seq_len}) < number of patches ({tokens})")
    if model_config.model_dim % model_config.num_head:
        raise ValueError("model_dim must be divisible by num_head")
    if adaptive:
        hs = model_config.channel_adaptivity_hidden_sizes
        if hs[2] % 2 or hs[2] // 2 != tokens:
            raise ValueError(
                f"channel_adaptivity_hidden_sizes[2] ({hs[2]}) must equal 2 x number of patches ({2 * tokens})")
        if model_config.adaptive_token_length != 6:
            raise ValueError("adaptive_token_length must be 6 (three encoders x two features per token)")


def _read_yaml(path: Path, what: str) -> dict:
    if not path.exists():
        raise FileNotFoundError(f"{what} configuration file not found: {path}")
    if path.suffix != ".yaml":
        raise ValueError(f"{what} configuration file must be a .yaml file: {path}")
    try:
        with open(path, "r") as fh:
            return yaml.safe_load(fh)
    except yaml.YAMLError as exc:
        raise ValueError(f"Failed to parse YAML file {path}: {exc}")


def load_config(system_config_path: Union[str, Path],
                model_config_path: Union[str, Path]) -> Tuple[SystemConfig, ModelConfig]:
    """YAML pair -> (SystemConfig, ModelConfig); ``ValueError`` on validation failure."""
    sys_path, mdl_path = Path(system_config_path), Path(model_config_path)
    raw_sys = _read_yaml(sys_path, "System")
    raw_mdl = _read_yaml(mdl_path, "Model")
    try:
        system_config = SystemConfig(**raw_sys)
    except ValidationError as exc:
        raise ValueError(f"System configuration validation for {sys_path} failed:\n{exc}")
    try:
        model_config = ModelConfig(**raw_mdl)
    except ValidationError as exc:
        raise ValueError(f"Model configuration validation for {mdl_path} failed:\n{exc}")
    _log.info("loaded configs %s , %s", sys_path, mdl_path)
    return system_config, model_config
