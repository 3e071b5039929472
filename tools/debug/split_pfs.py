import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch, bench
from adafortitran_amd import _abi, _lib, synth
from adafortitran_amd.hip_ops import engine_from_numpy, profile_kernel
c = bench.C3; B = 128; spec = bench._spec(c)
sd = synth.make_state_dict(**spec, adaptive_hidden=c["hidden"], max_seq_len=c["max_seq_len"], seed=bench.SEED)
inp = synth.make_inputs(B, seed=bench.SEED)
dev = lambda a: torch.from_numpy(a).to("cuda:0")
pil = dev(inp["pilots"]); meta = [dev(inp[k]) for k in ("snr", "ds", "dop")]
engs = {}
for name, path in (("pfs4", "adafortitran_amd/csrc/libaft_hip.so"), ("pfs6", "adafortitran_amd/csrc/libaft_hip_pfs6.so")):
    cfg = _abi.make_config(**spec, adaptive_hidden=c["hidden"]); cfg.precision = _abi.AFT_PRECISION_BF16X3
    engs[name] = engine_from_numpy(cfg, sd, "cuda:0", lib=_lib.load_path(os.path.abspath(path)))
    engs[name].forward(pil, *meta)
torch.cuda.synchronize()
def timed(fn, reps=20):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
res = {(n, k): [] for n in engs for k in ("chain", "forward")}
for rnd in range(7):
    for k in ("chain", "forward"):
        for n in (list(engs) if rnd % 2 == 0 else list(engs)[::-1]):
            e = engs[n]
            res[(n, k)].append(timed((lambda: profile_kernel(e, "chain", B, 1)) if k == "chain" else (lambda: e.forward(pil, *meta))))
for k in ("chain", "forward"):
    print(k, {n: round(statistics.median(res[(n, k)]), 1) for n in engs})
