"""adafortitran_amd -- MI355X (gfx950) native AdaFortiTran / FortiTran forward path.

Package contents (only what the hot path needs):
  csrc/          hand-written HIP kernels + the C ABI (include/adafortitran_amd.h)
  _lib, _abi     ctypes binding of that ABI (fails loudly when the .so is missing)
  hip_ops        pointer tables / workspace / per-stage calls on torch device memory
  estimators     drop-in mirror of the reference's src/models module surface
  blocks         parameter containers + the autograd (training) composite
  config         YAML/pydantic config surface
  synth          deterministic synthetic weights + inputs
  metrics        channel-MSE metric (device reduction + RCCL all-gather)
"""
from .config import ModelConfig, SystemConfig, load_config  # noqa: F401
from .estimators import (AdaFortiTranEstimator, BaseFortiTranEstimator, FortiTranEstimator,  # noqa: F401
                         LinearEstimator)

__version__ = "0.1.0"
