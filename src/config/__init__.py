from adafortitran_amd.config import load_config  # noqa: F401

__all__ = ["load_config"]
