#!/usr/bin/env python3
"""The split-precision tier (aft_config.precision = AFT_PRECISION_BF16X3) beside the exact-fp32 default, one process,
interleaved rounds:   python tools/ab_precision.py [--config C3|C2] [--batch B] [--rounds R] [--json OUT]

Rows: whole forward and the chain kernel classes; accuracy of the split tier against the fp32 path AND against the CPU
oracle on the first 8 frames (max|d| / |y|max, |dMSE| / MSE against the random target of the bench workload)."""
import argparse, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from adafortitran_amd import _abi, synth
from adafortitran_amd.hip_ops import engine_from_numpy, profile_kernel

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--json", default="")
args = ap.parse_args()
c = {"C3": bench.C3, "C2": bench.C2, "C5": bench.C5}[args.config]
B = args.batch or c["batch"]
spec = bench._spec(c)
sd = synth.make_state_dict(**spec, adaptive_hidden=c["hidden"], max_seq_len=c["max_seq_len"], seed=bench.SEED)
inp = synth.make_inputs(B, ofdm=c["ofdm"], pilot=c["pilot"], seed=bench.SEED)
dev = lambda a: torch.from_numpy(a).to("cuda:0")  # noqa: E731
pil = dev(inp["pilots"])
meta = [dev(inp[k]) for k in ("snr", "ds", "dop")] if c["hidden"] else [None] * 3
eng, out = {}, {}
for name, code in (("f32", _abi.AFT_PRECISION_F32), ("bf16x3", _abi.AFT_PRECISION_BF16X3)):
    cfg = _abi.make_config(**spec, adaptive_hidden=c["hidden"])
    cfg.precision = code
    eng[name] = engine_from_numpy(cfg, sd, "cuda:0")
    out[name] = torch.empty((B, *c["ofdm"]), dtype=torch.complex64, device="cuda:0")
    eng[name].forward(pil, *meta, out=out[name])
torch.cuda.synchronize()


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e3


rows = ["forward", "qkv", "chain", "chain_last"]
times = {(r, n): [] for r in rows for n in eng}
for rnd in range(args.rounds):
    for r in rows:
        order = list(eng)
        order = order[rnd % 2:] + order[:rnd % 2]
        for n in order:
            fn = (lambda n=n: eng[n].forward(pil, *meta, out=out[n])) if r == "forward" else (lambda n=n, r=r: profile_kernel(eng[n], r, B, 1))
            times[(r, n)].append(timed(fn))
for n in eng:
    eng[n].forward(pil, *meta, out=out[n])
torch.cuda.synchronize()
a, b = out["f32"].cpu().numpy(), out["bf16x3"].cpu().numpy()
tgt = inp["target"]
mse = lambda y: float(np.mean(np.abs(y - tgt) ** 2, dtype=np.float64))  # noqa: E731
res = {"config": args.config, "batch": B, "rows": {},
       "bf16x3_vs_f32": {"max_abs_over_ymax": float(np.abs(a - b).max() / np.abs(a).max()), "rel_dMSE": abs(mse(b) - mse(a)) / mse(a)}}
try:
    from oracle import oracle
    k = 8
    oargs = [inp[x][:k] for x in ("snr", "ds", "dop")] if c["hidden"] else [None] * 3
    ref = oracle.Oracle(eng["f32"].cfg, sd).forward(inp["pilots"][:k], *oargs)
    for n, y in (("f32", a), ("bf16x3", b)):
        m_ref, m_y = float(np.mean(np.abs(ref - tgt[:k]) ** 2, dtype=np.float64)), float(np.mean(np.abs(y[:k] - tgt[:k]) ** 2, dtype=np.float64))
        res[n + "_vs_oracle"] = {"max_abs_over_ymax": float(np.abs(y[:k] - ref).max() / np.abs(ref).max()), "rel_dMSE": abs(m_y - m_ref) / m_ref}
except Exception as exc:
    res["oracle_error"] = str(exc)[:100]
print(f"config {args.config} B={B}; us: median (min) over {args.rounds} interleaved rounds of {args.reps}")
for r in rows:
    med = {n: statistics.median(times[(r, n)]) for n in eng}
    print(f"{r:<11}" + "".join(f"{n:>8}: {med[n]:8.1f} ({min(times[(r, n)]):8.1f})" for n in eng) + f"   ratio {med['f32'] / med['bf16x3']:.2f}x")
    res["rows"][r] = {n: round(med[n], 2) for n in eng}
res["frames_per_s"] = {n: round(B / res["rows"]["forward"][n] * 1e6, 1) for n in eng}
print(json.dumps({k: v for k, v in res.items() if k != "rows"}, indent=1))
if args.json:
    json.dump(res, open(args.json, "w"), indent=1)
