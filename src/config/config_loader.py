from adafortitran_amd.config import load_config  # noqa: F401
