"""Import-compatibility shim: lets the reference's trainer/eval scaffolding keep its
``from src.models import ...`` / ``from src.config import load_config`` lines while the
implementation lives in ``adafortitran_amd`` (gfx950 HIP kernels)."""
