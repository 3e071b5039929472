#!/usr/bin/env python3
"""A/B of the two encoder paths of ONE library build, interleaved rounds in one process (guide rule 24):

    python tools/ab_encoder.py [--config C3|C2] [--batch B] [--rounds R] [--reps N] [--json OUT]

  launches : embed+QKV, [attention, chain] x L  -- 13 launches at L = 6 (k_chain.hip, k_attn.hip)
  plane    : the plane-resident encoder kernel, one launch (k_encoder.hip)

Rows: whole forward (aft_forward_f32 with cfg.encoder_path forced) and the encoder alone (the plane kernel through
aft_profile_kernel_f32 against the launch path's kernels replayed in forward order between one event pair).  Also prints
whether the two forwards give identical bits."""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from adafortitran_amd import _abi, synth  # noqa: E402
from adafortitran_amd.hip_ops import engine_from_numpy, profile_kernel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--json", default="")
args = ap.parse_args()
c = {"C3": bench.C3, "C2": bench.C2}[args.config]
B = args.batch or c["batch"]
spec = bench._spec(c)
sd = synth.make_state_dict(**spec, adaptive_hidden=c["hidden"], max_seq_len=c["max_seq_len"], seed=bench.SEED)
inp = synth.make_inputs(B, ofdm=c["ofdm"], pilot=c["pilot"], seed=bench.SEED)
dev = lambda a: torch.from_numpy(a).to("cuda:0")  # noqa: E731
pil = dev(inp["pilots"])
meta = [dev(inp[k]) for k in ("snr", "ds", "dop")] if c["hidden"] else [None] * 3
L = c["num_layers"]
paths = {"launches": _abi.AFT_ENCODER_LAUNCHES, "plane": _abi.AFT_ENCODER_PLANE}
eng, out = {}, {}
for name, code in paths.items():
    cfg = _abi.make_config(**spec, adaptive_hidden=c["hidden"])
    cfg.encoder_path = code
    eng[name] = engine_from_numpy(cfg, sd, "cuda:0")
    out[name] = torch.empty((B, *c["ofdm"]), dtype=torch.complex64, device="cuda:0")
    eng[name].forward(pil, *meta, out=out[name])
torch.cuda.synchronize()
same = bool(torch.equal(torch.view_as_real(out["launches"]), torch.view_as_real(out["plane"])))


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e3


def encoder_launches():
    e = eng["launches"]
    profile_kernel(e, "qkv", B, 1)
    for _ in range(L - 1):
        profile_kernel(e, "attention", B, 1)
        profile_kernel(e, "chain", B, 1)
    profile_kernel(e, "attention", B, 1)
    profile_kernel(e, "chain_last", B, 1)


rows = {"forward": {n: (lambda n=n: eng[n].forward(pil, *meta, out=out[n])) for n in paths},
        "encoder": {"launches": encoder_launches, "plane": lambda: profile_kernel(eng["plane"], "encoder_plane", B, 1)}}
times = {(r, n): [] for r in rows for n in paths}
for rnd in range(args.rounds):
    for r, fns in rows.items():
        order = list(paths)
        order = order[rnd % 2:] + order[:rnd % 2]
        for n in order:
            times[(r, n)].append(timed(fns[n]))
fl = bench.algorithmic_flops(c, B)
print(f"config {args.config} B={B} ({2 * B} planes); us: median (min) over {args.rounds} interleaved rounds of {args.reps}")
res = {"config": args.config, "batch": B, "bit_identical": same, "rows": {}}
for r in rows:
    med = {n: statistics.median(times[(r, n)]) for n in paths}
    mn = {n: min(times[(r, n)]) for n in paths}
    print(f"{r:<10}" + "".join(f"{n:>10}: {med[n]:8.1f} ({mn[n]:8.1f})" for n in paths) + f"   plane/launches = {med['plane'] / med['launches']:.4f}")
    res["rows"][r] = {n: {"median_us": round(med[n], 2), "min_us": round(mn[n], 2)} for n in paths}
    if r == "encoder":
        for n in paths:
            res["rows"][r][n]["mfma_util"] = round(fl["encoder_total"] / med[n] / 1e6 / bench.PEAK_FP32_MFMA_TFLOPS, 4)
            print(f"   encoder MFMA utilisation ({n}): {res['rows'][r][n]['mfma_util']:.4f} of {bench.PEAK_FP32_MFMA_TFLOPS} TFLOP/s")
print("forward outputs bit-identical:", same)
if args.json:
    with open(args.json, "w") as fh:
        json.dump(res, fh, indent=1)
