from adafortitran_amd.blocks import *  # noqa: F401,F403
from adafortitran_amd.blocks import __all__  # noqa: F401
