#!/usr/bin/env python3
"""A/B timing of library builds in ONE process, interleaved rounds (guide rule 24):

    python tools/ab_kernels.py [--config C3|C2|C5] [--batch B] [--rounds R] [--kernels chain,attention,...] \
           base=adafortitran_amd/csrc/libaft_hip.so v1=adafortitran_amd/csrc/libaft_hip_v1.so ...

Per build and kernel class: median and min over the rounds of the average launch time (20 launches between one event
pair, kernels replayed through aft_profile_kernel_f32 on the activations of a real forward), plus the whole-forward time,
and max|out - out_base| of the forward output so a faster-but-different variant is visible immediately."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from adafortitran_amd import _abi, _lib, synth  # noqa: E402
from adafortitran_amd.hip_ops import engine_from_numpy, profile_kernel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--kernels", default="upsample,embed,qkv,attention,chain,chain_last,tail,forward")
ap.add_argument("libs", nargs="+")
args = ap.parse_args()
c = {"C3": bench.C3, "C2": bench.C2, "C5": bench.C5}[args.config]
B = args.batch or c["batch"]
spec = bench._spec(c)
sd = synth.make_state_dict(**spec, adaptive_hidden=c["hidden"], max_seq_len=c["max_seq_len"], seed=bench.SEED)
cfg = _abi.make_config(**spec, adaptive_hidden=c["hidden"])
inp = synth.make_inputs(B, ofdm=c["ofdm"], pilot=c["pilot"], seed=bench.SEED)
dev = lambda a: torch.from_numpy(a).to("cuda:0")  # noqa: E731
pil = dev(inp["pilots"])
meta = [dev(inp[k]) for k in ("snr", "ds", "dop")] if c["hidden"] else [None] * 3
names, engines, outs = [], [], []
for item in args.libs:
    name, path = item.split("=", 1) if "=" in item else (os.path.basename(item), item)
    eng = engine_from_numpy(cfg, sd, "cuda:0", lib=_lib.load_path(os.path.abspath(path)))
    out = torch.empty((B, *c["ofdm"]), dtype=torch.complex64, device="cuda:0")
    eng.forward(pil, *meta, out=out)
    torch.cuda.synchronize()
    names.append(name); engines.append(eng); outs.append(out)
kernels = args.kernels.split(",")
times = {(n, k): [] for n in names for k in kernels}


def one(eng, out, k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if k == "forward":
        e0.record()
        for _ in range(args.reps):
            eng.forward(pil, *meta, out=out)
        e1.record()
    else:
        io = pil if k == "upsample" else (out if k == "tail" else None)
        profile_kernel(eng, k, B, 2, io)
        e0.record()
        profile_kernel(eng, k, B, args.reps, io)
        e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e3


for rnd in range(args.rounds):
    for k in kernels:
        # the build measured first after a switch of kernel class ran ~2 us slow in the whole-forward rows (observed
        # with the SAME library listed twice): rotate who goes first, and throw one launch sequence away before timing
        order = list(zip(names, engines, outs))
        order = order[rnd % len(order):] + order[:rnd % len(order)]
        one(order[0][1], order[0][2], k)
        for n, eng, out in order:
            times[(n, k)].append(one(eng, out, k))
for n, eng, out in zip(names, engines, outs):      # leave every `out` holding a forward result
    eng.forward(pil, *meta, out=out)
torch.cuda.synchronize()
print(f"config {args.config} B={B}; us per launch: median (min) over {args.rounds} interleaved rounds of {args.reps} launches")
print(f"{'kernel':<12}" + "".join(f"{n:>22}" for n in names))
for k in kernels:
    print(f"{k:<12}" + "".join(f"{statistics.median(times[(n, k)]):>13.1f} ({min(times[(n, k)]):>6.1f})" for n in names))
ref = torch.view_as_real(outs[0])
for n, out in zip(names[1:], outs[1:]):
    d = (torch.view_as_real(out) - ref).abs().max().item()
    print(f"max|out[{n}] - out[{names[0]}]| = {d:.3e}   (|y|max {ref.abs().max().item():.3f})")
