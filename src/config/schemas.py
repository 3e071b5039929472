from adafortitran_amd.config import BaseConfig, ModelConfig, OFDMParams, PilotParams, SystemConfig  # noqa: F401
