"""Round 6: does aft_debug_fill_lds_f32 reach a kernel's uninitialised LDS reads?  A variant built with -DAFT_TEST_LDS_BUG=1
(python -m adafortitran_amd.build --variant ldsbug -DAFT_TEST_LDS_BUG=1) re-opens round 6's embed_any_kernel bug; its general-engine
forward must come out clean in a fresh process and NaN behind a NaN fill."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ctypes as C
import numpy as np, torch
from adafortitran_amd import _abi, _lib, synth
from adafortitran_amd.hip_ops import engine_from_numpy
lib = _lib.load_path(os.path.join(os.path.dirname(_lib.lib_path()), "libaft_hip_ldsbug.so"))
lib.aft_set_switch(b"AFT_EMBED_ANY_OLD", b"1")   # the kernel the bug lived in (since late round 6 the product runs k_ends_train.hip's)
spec = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=2, model_dim=512, num_head=8)
sd = synth.make_state_dict(**spec, adaptive_hidden=(7, 42, 560), seed=3)
cfg = _abi.make_config(**spec, adaptive_hidden=(7, 42, 560))
eng = engine_from_numpy(cfg, sd, "cuda:0", lib=lib)
inp = synth.make_inputs(4, seed=4)
dev = lambda a: torch.from_numpy(a).cuda()
args = (dev(inp["pilots"]), dev(inp["snr"]), dev(inp["ds"]), dev(inp["dop"]))
out0 = eng.forward(*args).clone()
print("fresh process: finite", bool(torch.isfinite(torch.view_as_real(out0)).all()))
for v in (float("nan"), 1e30):
    _lib.check(lib.aft_debug_fill_lds_f32(C.c_float(v), _lib.current_stream_ptr(torch.device("cuda:0"))))
    out = eng.forward(*args)
    print("behind a fill of", v, ": finite", bool(torch.isfinite(torch.view_as_real(out)).all()), "same bits", bool(torch.equal(torch.view_as_real(out), torch.view_as_real(out0))))
