#!/usr/bin/env python3
"""bench.py -- OFDM frames/s of the AdaFortiTran forward path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C2|C5] [--batch B]

``--gpus N`` with N > 1 and no torchrun environment: this process touches no GPU, starts
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` on
itself, relays rank 0's JSON line and exits with the children's status (fewer than N visible
devices => non-zero exit).  Under torchrun (WORLD_SIZE set) it is one rank: one process per GPU.

A *step* is one pass of the hot path over one batch of synthetic input on every rank, in the form SURVEY.md
8(d) defines the metric: ``model(pilots_cpu, meta_cpu)`` through the module surface exactly as the
reference's evaluator calls it (trainer.py:328-347) -- pilots + meta arrive as CPU tensors, their transfer
is INSIDE the step, the estimate stays on the device -- plus the device-side channel-MSE partial sum.  That
is ``value`` (= ``value_h2d_inclusive``); the rate of the bare engine call with the inputs already resident
in HBM rides along as ``value_resident`` (0.5-1 % higher: the kernels read the pinned staging slot directly,
no copy is enqueued).
``--config`` picks the per-rank workload: C3 = AdaFortiTran default, 128 frames per GPU (the headline;
``--gpus 8`` makes it BASELINE config 4), C2 = FortiTran default, C5 = 240x28 / 12 layers / d = 256 at
64 frames per GPU (``--gpus 8`` = BASELINE config 5: batch 512 over 8 GPUs).  Frames shard across ranks
with no data-path collective (weak scaling); ONE RCCL all-gather of the per-rank (sum|e|^2, n) pair
closes the sweep.  Rank 0 prints ONE compact JSON line on stdout (numbers, short keys:
DESIGN.md section 5 explains every field); ``--verbose-json PATH`` also writes the annotated record.

Fields besides the contract: roofline (dominant kernel, live event timing) . kernels (ms / TFLOP/s per
class) . upsampler (graded fraction (ii) of SURVEY 8d and the by-construction HBM fraction (i)) .
encoder_mfma_util . per_rank (N > 1: step ms min/max over ranks, all-gather latency) . and at N = 1:
parity (vs the oracle, bounded sample) . configs (C1, the other default model, C5) . next_rows (SURVEY 8f:
training step, ingest, evaluation sweep, LS baseline) . split_precision (the opt-in bf16x3 tier, reported
separately, never `value`) . cpu_baseline (the reference-equivalent CPU path on this box's host cores).

``--stub`` (tests only, tests/test_bench_cli.py): the N-rank control flow with the GPU work replaced by
CPU sleeps over gloo -- prints ``"data": "stub"`` and measures nothing.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4, dense, = fp32 vector peak
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
SEED = 20251114
PREWARM_STEPS = 300             # untimed device spin-up before the W warm-up steps (~0.5 s; reported as `prewarm_steps`)

C3 = dict(name="C3", model="adafortitran", ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=128,
          num_head=4, hidden=(7, 42, 560), max_seq_len=512, batch=128,
          label="AdaFortiTran default config (6 layers, d=128, 4 heads, adaptive tokens), 120x14 grid, pilots 12x2")
C2 = dict(C3, name="C2", model="fortitran", hidden=None,
          label="FortiTran default config (6 layers, d=128, 4 heads), 120x14 grid, pilots 12x2")
C5 = dict(name="C5", model="adafortitran", ofdm=(240, 28), pilot=(24, 4), patch=(3, 2), num_layers=12, model_dim=256,
          num_head=8, hidden=(7, 42, 2240), max_seq_len=1120, batch=64,
          label="AdaFortiTran 12 layers / d=256 / 8 heads, 240x28 grid, patch [3,2], pilots 24x4, 64 frames per GPU "
                "(batch 512 over 8 GPUs)")
CONFIGS = {"C3": C3, "C2": C2, "C5": C5}


def _spec(c):
    return dict(ofdm=c["ofdm"], pilot=c["pilot"], patch=c["patch"], num_layers=c["num_layers"], model_dim=c["model_dim"],
                num_head=c["num_head"])


def algorithmic_flops(c, batch):
    """FLOPs per launch (MAC = 2) of each kernel class: SURVEY.md 2.2 / 8(d) per-plane figures x the planes one
    launch processes (DESIGN.md 'roofline accounting')."""
    (S, T), (Ps, Pt), (p0, p1) = c["ofdm"], c["pilot"], c["patch"]
    d, L, planes = c["model_dim"], c["num_layers"], 2 * batch
    tokens = (S // p0) * (T // p1)
    rows = planes * tokens
    qkv = 2 * rows * d * 3 * d
    proj = 2 * rows * d * d
    ffn = 2 * rows * d * 2 * d * 2
    attn = 2 * 2 * planes * tokens * tokens * d          # QK^T + PV over all heads
    conv = planes * S * T * 2 * 9 * (1 * 8 + 8 * 32 + 32 * 8 + 8 * 1)
    up = planes * 2 * Ps * Pt * S * T
    pin = p0 * p1 + (6 if c["hidden"] else 0)
    emb, lin2 = 2 * rows * d * pin, 2 * rows * d * p0 * p1
    # the first chain launch also does patch embedding + linear_1, the last one linear_2 (fused since round 2)
    # the pilot_upsampler product (`up`, 0.5 % of the stage's FLOPs) runs in the forward's prologue launch since round 4 (one product over
    # all planes): the conv-head launch that the "upsample" class times is the initial ConvEnhancer alone
    fl = {"qkv": qkv + emb, "chain": proj + ffn + qkv, "chain_last": proj + ffn + lin2, "attention": attn, "upsample": conv,
          "tail": conv, "encoder_total": L * (qkv + proj + ffn + attn)}
    fl["up_product"] = up
    fl["forward_total"] = fl["upsample"] + up + emb + fl["encoder_total"] + lin2 + fl["tail"]
    return fl


def upsampler_bytes_per_frame(c):
    """Compulsory HBM bytes per frame of the upsampler stage: pilots in, conv_enhanced (2 planes) out."""
    return c["pilot"][0] * c["pilot"][1] * 8 + 2 * c["ofdm"][0] * c["ofdm"][1] * 4


def host_cpu_info():
    """CPU model, sockets, physical cores, logical CPUs ('print CPU model, physical core count')."""
    model, pairs, sockets = "unknown", set(), set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                    sockets.add(phys)
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return {"cpu_model": model, "sockets": max(1, len(sockets)), "physical_cores": len(pairs) or logical,
            "logical_cpus": logical}


def one_socket_cores():
    """One logical CPU per physical core of the socket that holds CPU 0's neighbours (/proc/cpuinfo), [] if unknown."""
    by_core, cur = {}, {}
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if ":" in line:
                    k, v = (t.strip() for t in line.split(":", 1))
                    cur[k] = v
                elif not line.strip() and cur:
                    if "processor" in cur and "physical id" in cur and "core id" in cur:
                        by_core.setdefault((cur["physical id"], cur["core id"]), int(cur["processor"]))
                    cur = {}
    except OSError:
        return []
    if not by_core:
        return []
    first = sorted(k[0] for k in by_core)[0]
    return sorted(v for k, v in by_core.items() if k[0] == first)


# ------------------------------------------------------------------------------------------------------------------
# self-launch (N > 1 without torchrun): nothing above or inside touches the GPU
# ------------------------------------------------------------------------------------------------------------------
def self_launch(args, argv) -> int:
    if not args.stub and not args.share_gpu:
        import torch   # device_count() does not initialise the GPU on this image
        n = torch.cuda.device_count()
        if n < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {n} device(s) visible", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if proc.returncode != 0 or line is None:
        sys.stdout.write(proc.stdout)
        print(f"bench.py: the {args.gpus}-rank launch failed (exit {proc.returncode})", file=sys.stderr)
        return proc.returncode or 3
    print(line)
    return 0


# ------------------------------------------------------------------------------------------------------------------
# workloads (one rank)
# ------------------------------------------------------------------------------------------------------------------
class Workload:
    """Engine + synthetic inputs of one config, resident on `device`."""

    def __init__(self, c, device, rank=0, batch=None):
        import torch
        from adafortitran_amd import _abi, synth
        from adafortitran_amd.hip_ops import engine_from_numpy
        from adafortitran_amd.metrics import MseAccumulator
        self.c, self.device = c, device
        self.B = batch or c["batch"]
        self.adaptive = c["hidden"] is not None
        self.sd = synth.make_state_dict(**_spec(c), adaptive_hidden=c["hidden"], max_seq_len=c["max_seq_len"], seed=SEED)
        self.cfg = _abi.make_config(**_spec(c), adaptive_hidden=c["hidden"])
        self.eng = engine_from_numpy(self.cfg, self.sd, device)
        self.inp = synth.make_inputs(self.B, ofdm=c["ofdm"], pilot=c["pilot"], seed=SEED + 1000 * rank)
        self.pil = torch.from_numpy(self.inp["pilots"]).to(device)
        self.tgt = torch.from_numpy(self.inp["target"]).to(device)
        self.meta = [torch.from_numpy(self.inp[k]).to(device) for k in ("snr", "ds", "dop")] if self.adaptive else [None] * 3
        self.out = torch.empty((self.B, *c["ofdm"]), dtype=torch.complex64, device=device)
        self.acc = MseAccumulator(device)
        self.sync = torch.cuda.synchronize
        self.model = None

    def surface_step(self):
        """The metric's step (reference trainer.py:328-347): the module called with CPU pilots + the collated CPU meta
        tuple, H2D inside the step, estimate left on the device, device-side MSE partial."""
        import torch
        if self.model is None:
            from adafortitran_amd import synth
            self.model = make_module(self.c, str(self.device), self.sd)
            self.pil_cpu = torch.from_numpy(self.inp["pilots"])
            self.meta_cpu = synth.meta_tuple(self.inp) if self.adaptive else None
        with torch.no_grad():
            est = self.model(self.pil_cpu, self.meta_cpu) if self.adaptive else self.model(self.pil_cpu)
        self.acc.update(est, self.tgt)
        return est

    def forward(self):
        # the stateless entry point, exactly what the module surface calls in eval mode (estimators.py): the encoder weights are
        # re-laid into fragment order inside every call, in the channel adapter's launch (nothing cached that could go stale)
        return self.eng.forward(self.pil, *self.meta, out=self.out)

    def step(self):
        self.forward()
        self.acc.update(self.out, self.tgt)          # device-side partial sum, no host sync

    def clock(self):
        import torch
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    @staticmethod
    def elapsed_ms(a, b):
        return a.elapsed_time(b)


class StubWorkload:
    """tests/test_bench_cli.py only: the control flow of a rank (warm-up, fences, timed steps, metric all-gather,
    rank-0-only legs, final barrier) with the GPU work replaced by a short host sleep; CPU tensors, gloo."""

    def __init__(self, c, device, rank=0, batch=None):
        import torch
        from adafortitran_amd.metrics import MseAccumulator
        self.c, self.device, self.B = c, torch.device("cpu"), batch or c["batch"]
        self.adaptive = c["hidden"] is not None
        g = torch.Generator().manual_seed(SEED + 1000 * rank)
        shape = (self.B, 4, 4)                          # the stub's "estimates": tiny, so that the sleeps dominate a step
        self.out = torch.complex(torch.randn(shape, generator=g), torch.randn(shape, generator=g))
        self.tgt = torch.complex(torch.randn(shape, generator=g), torch.randn(shape, generator=g))
        self.acc = MseAccumulator("cpu")
        self.sync = lambda: None
        self.sleep = 0.004 * (1 + rank)                 # ranks differ: per-rank min/max must show it

    def surface_step(self):
        self.step()

    def forward(self):
        time.sleep(self.sleep)
        return self.out

    def step(self):
        self.forward()
        self.acc.update(self.out, self.tgt)

    def clock(self):
        return time.perf_counter()

    @staticmethod
    def elapsed_ms(a, b):
        return (b - a) * 1e3


def rank_identity(rank, local_rank, device, args):
    """What this rank ran on, for the N-rank record (bench line `per_rank.ranks`)."""
    import torch
    ident = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "pid": os.getpid()}
    if device.type != "cuda":
        ident.update({"device": "cpu", "pci": f"cpu:{os.getpid()}"})
        return ident
    props = torch.cuda.get_device_properties(device)
    dom, bus, dev = (getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    uuid = getattr(props, "uuid", None)
    pci = f"{dom:04x}:{bus:02x}:{dev:02x}" if None not in (dom, bus, dev) else (str(uuid) if uuid is not None else f"index{device.index}")
    ident.update({"device": device.index, "pci": pci, "pci_known": None not in (dom, bus, dev) or uuid is not None, "name": props.name,
                  "visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", ""))})
    return ident


def collective_lib_version(args):
    """RCCL's version as torch reports it (backend "nccl" IS RCCL on ROCm); gloo has none."""
    if args.stub or args.share_gpu:
        return "gloo"
    try:
        import torch
        return "rccl " + ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as exc:
        return f"unknown ({type(exc).__name__})"


def timed_steps(wl, step, steps, warmup, fence):
    """W untimed + exactly K timed steps between fences; returns (wall s, device ms, per-step device ms sorted)."""
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    marks = [wl.clock()]
    for _ in range(steps):
        step()
        marks.append(wl.clock())
    fence()
    wall = time.perf_counter() - t0
    per_step = sorted(wl.elapsed_ms(a, b) for a, b in zip(marks[:-1], marks[1:]))
    return wall, wl.elapsed_ms(marks[0], marks[-1]), per_step


def kernel_times(wl, reps):
    """Average duration (ms) of every kernel class inside a real forward.

    Pass 1: `reps` REAL forwards (aft_forward_f32) between ONE event pair: T_fwd, free of per-launch event overhead.
    Pass 2: the kernel classes in the launch order of the forward (prologue, conv head, embed+QKV, [attention, chain] x (L-1),
    attention, last chain + linear_2, conv tail -- same kernels / grids / arguments through aft_profile_kernel_f32)
    with an event pair around every launch.  A pair inflates its launch by a fixed few us: the excess of the raw sum
    over T_fwd, divided by the number of launches, is subtracted from every launch.  Events are recorded on torch's
    current stream = the stream the library launches on."""
    import torch
    from adafortitran_amd.hip_ops import profile_kernel
    L = wl.c["num_layers"]
    # the prologue hook takes [pilots | snr | ds | dop] as one buffer (include/adafortitran_amd.h, AFT_KERNEL_PROLOGUE)
    parts = [torch.view_as_real(wl.pil).reshape(-1)] + [(m.reshape(-1).float() if m is not None else torch.zeros(wl.B, device=wl.device)) for m in wl.meta]
    pro_in = torch.cat(parts).contiguous()
    flow = [("prologue", pro_in), ("upsample", wl.pil), ("qkv", None)]
    for _ in range(L - 1):
        flow += [("attention", None), ("chain", None)]
    flow += [("attention", None), ("chain_last", None), ("tail", wl.out)]
    launches = {}
    for which, _ in flow:
        launches[which] = launches.get(which, 0) + 1
    for _ in range(100):       # the device idled while rank 0 collected the headline: back to sustained clocks first (0.16 s);
        wl.forward()            # also fills the workspace the profile hook replays on
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        wl.forward()
    e1.record()
    torch.cuda.synchronize()
    t_fwd = e0.elapsed_time(e1) / reps
    for _ in range(20):
        for which, io in flow:
            profile_kernel(wl.eng, which, wl.B, 1, io)
    pairs = []
    for _ in range(reps):
        for which, io in flow:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            profile_kernel(wl.eng, which, wl.B, 1, io)
            b.record()
            pairs.append((which, a, b))
    torch.cuda.synchronize()
    tot = {}
    for which, a, b in pairs:
        tot[which] = tot.get(which, 0.0) + a.elapsed_time(b)
    raw = {k: tot[k] / reps / launches[k] for k in tot}
    # An event pair costs its launch a FIXED few microseconds (the same for a 12-us and a 150-us kernel), and the flow is the whole
    # forward, launch for launch: whatever the raw times sum to beyond T_fwd is that overhead, n_launches times -- subtracted per
    # launch (round 3 scaled proportionally, which charged the short kernels too little and the conv stacks 10 % too much against
    # the rocprofv3 trace of the same command)
    n_launch = sum(launches.values())
    overhead = max(0.0, (sum(raw[k] * launches[k] for k in raw) - t_fwd) / n_launch)
    ms = {k: max(raw[k] - overhead, 0.5 * raw[k]) for k in raw}
    share, pro_with, pro_without = prologue_product_share_ms(wl, pro_in)
    ms["_up_product_share"], ms["_prologue_back_to_back"], ms["_prologue_without_product"] = share, pro_with, pro_without
    return ms, raw, t_fwd


def prologue_product_share_ms(wl, pro_in, reps=30):
    """Time of the forward's prologue launch WITH and WITHOUT the pilot_upsampler product (K1 of the upsampler stage, SURVEY 8(d):
    the stage is K0 + K1 + K2 and round 4 moved K1 into this launch): interleaved rounds of `reps` launches between one event pair
    each, median over rounds; AFT_PROLOGUE_NO_UP is the library's measurement switch (aft_profile_kernel_f32 only).  The difference
    is what the product costs the forward and is charged to the upsampler stage."""
    import numpy as np
    import torch
    from adafortitran_amd import _lib
    from adafortitran_amd.hip_ops import profile_kernel
    res = {"with": [], "without": []}
    try:
        for _ in range(5):
            for key in ("with", "without"):
                _lib.set_switch("AFT_PROLOGUE_NO_UP", "1" if key == "without" else None)
                profile_kernel(wl.eng, "prologue", wl.B, 3, pro_in)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                profile_kernel(wl.eng, "prologue", wl.B, reps, pro_in)
                e1.record()
                e1.synchronize()
                res[key].append(e0.elapsed_time(e1) / reps)
    finally:
        _lib.set_switch("AFT_PROLOGUE_NO_UP", None)
    w, wo = float(np.median(res["with"])), float(np.median(res["without"]))
    return max(w - wo, 0.0), w, wo


def kernel_report(wl, reps):
    """(kernels, roofline of the dominant kernel, encoder MFMA utilisation, forward ms, flops, upsampler record)"""
    c, B = wl.c, wl.B
    fl = algorithmic_flops(c, B)
    with unsplit():          # per-kernel accounting is defined on the one-lane launch sequence (a no-op at the stated batches)
        ms, raw, t_flow = kernel_times(wl, reps)
    L = c["num_layers"]
    share = ms.pop("_up_product_share")
    pro_with, pro_without = ms.pop("_prologue_back_to_back"), ms.pop("_prologue_without_product")
    kernels = {k: ({"ms": round(v, 4), "tflops": round(fl[k] / v / 1e9, 2)} if k in fl else {"ms": round(v, 4)}) for k, v in ms.items()}
    enc_ms = ms["qkv"] + L * ms["attention"] + (L - 1) * ms["chain"] + ms["chain_last"]
    dom = max(("chain", "attention"), key=lambda k: ms[k] * ((L - 1) if k == "chain" else L))
    names = {"chain": f"chain_kernel<{c['model_dim']},GELU,MLP,QKV>", "attention": "attn_kernel"}
    achieved = fl[dom] / ms[dom] / 1e9
    roof = {"kernel": names[dom], "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "flops_per_launch": fl[dom],
            "ms_per_launch": round(ms[dom], 4), "launches_per_forward": (L - 1) if dom == "chain" else L}
    # upsampler stage: graded fraction (ii) = achieved FLOP/s / min(peak, AI x BW) -- fused, its arithmetic
    # intensity (2 354 FLOP/B) puts the ridge far above the fp32 peak, so the roof is the fp32 MFMA peak; (i) = compulsory
    # bytes x frames/s / HBM peak (tiny by construction once fused)
    # The stage is counted WHOLE: FLOPs = pilot_upsampler product (K1) + initial ConvEnhancer (K2); time = the conv-head
    # launch + what the product adds to the prologue launch (measured: prologue with minus without it).  `conv_only_frac` is the
    # conv-head launch alone (round 4's `frac`).  The tail stage (fold + residual + final ConvEnhancer) is one launch: `tail_frac`.
    ub = upsampler_bytes_per_frame(c)
    stage_ms = ms["upsample"] + share
    up_tf = (fl["upsample"] + fl["up_product"]) / stage_ms / 1e9
    ai = (fl["upsample"] + fl["up_product"]) / (ub * B)
    upsampler = {"tflops": round(up_tf, 2), "frac": round(up_tf / min(PEAK_FP32_MFMA_TFLOPS, ai * PEAK_HBM_GBS / 1e3), 4),
                 "stage_ms": round(stage_ms, 4), "conv_head_ms": round(ms["upsample"], 4), "product_share_of_prologue_ms": round(share, 4),
                 "prologue_ms_with_without_product": [round(pro_with, 4), round(pro_without, 4)],
                 "conv_only_frac": round(fl["upsample"] / ms["upsample"] / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                 "ai_flop_per_byte": round(ai), "hbm_frac": round(ub * B / (stage_ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 6),
                 "tail_frac": round(fl["tail"] / ms["tail"] / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4)}
    return kernels, roof, round(fl["encoder_total"] / enc_ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4), round(t_flow, 4), fl, upsampler


def parity_vs_oracle(wl, sample):
    """HIP path vs the CPU oracle on the first `sample` frames of the workload's own inputs (oracle = checker only)."""
    import numpy as np
    from oracle import oracle
    idx = slice(0, sample)
    args = [wl.inp[k][idx] for k in ("snr", "ds", "dop")] if wl.adaptive else [None] * 3
    ref = oracle.Oracle(wl.cfg, wl.sd).forward(wl.inp["pilots"][idx], *args)
    got = wl.forward()[idx].cpu().numpy()
    tgt = wl.inp["target"][idx]
    mse_hip = float(np.mean(np.abs(got - tgt) ** 2, dtype=np.float64))
    mse_ref = float(np.mean(np.abs(ref - tgt) ** 2, dtype=np.float64))
    return {"frames": sample, "max_abs": float(np.abs(got - ref).max()), "ymax": float(np.abs(ref).max()),
            "mean_sq": float(np.mean(np.abs(got - ref) ** 2, dtype=np.float64)),
            "rel_dMSE": abs(mse_hip - mse_ref) / mse_ref, "tol_max_abs_over_ymax": 5e-5, "tol_rel_dMSE": 1e-4}


def make_module(c, device_str, sd):
    import torch
    import adafortitran_amd as A
    adaptive = c["hidden"] is not None
    sc = A.SystemConfig(ofdm=dict(num_scs=c["ofdm"][0], num_symbols=c["ofdm"][1]),
                        pilot=dict(num_scs=c["pilot"][0], num_symbols=c["pilot"][1]))
    kw = dict(model_type=c["model"], patch_size=tuple(c["patch"]), num_layers=c["num_layers"], model_dim=c["model_dim"],
              num_head=c["num_head"], max_seq_len=c["max_seq_len"], device=device_str)
    if adaptive:
        kw.update(channel_adaptivity_hidden_sizes=list(c["hidden"]), adaptive_token_length=6)
    model = (A.AdaFortiTranEstimator if adaptive else A.FortiTranEstimator)(sc, A.ModelConfig(**kw)).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return model


def module_surface(wl, steps, warmup, surface_fps):
    """The bare engine call on inputs resident in HBM beside the headline's module-surface step (SURVEY.md 8(d) form: model(pilots_cpu,
    meta_cpu) as the reference's evaluator calls it, trainer.py:332-337): same step count right behind the headline, then both over a
    long run at the same moment, and the bit identity of the two outputs."""
    import torch
    wall_res0, _, _ = timed_steps(wl, wl.step, steps, warmup, torch.cuda.synchronize)
    long_steps = max(200, steps)
    wall_long, _, _ = timed_steps(wl, wl.surface_step, long_steps, 5, torch.cuda.synchronize)
    wall_res, _, _ = timed_steps(wl, wl.step, long_steps, 5, torch.cuda.synchronize)     # resident inputs, same moment, same length
    est = wl.surface_step()
    same = bool(torch.equal(torch.view_as_real(est), torch.view_as_real(wl.forward())))
    fps_res0 = wl.B * steps / wall_res0
    fps_long, fps_res = wl.B * long_steps / wall_long, wl.B * long_steps / wall_res
    return {"value": round(surface_fps, 1), "resident_value": round(fps_res0, 1), "ratio_to_resident": round(surface_fps / fps_res0, 4),
            "long_run": {"steps": long_steps, "value": round(fps_long, 1), "resident_value": round(fps_res, 1),
                         "ratio_to_resident": round(fps_long / fps_res, 4)},
            "h2d": "none enqueued: the kernels read the pinned ring slot directly", "bit_identical_to_engine": same}


def config_record(c, device, steps, warmup, oracle_sample, kernel_reps):
    import torch
    wl = Workload(c, device)
    wall, dev_ms, per = timed_steps(wl, wl.surface_step, steps, warmup, torch.cuda.synchronize)   # the metric's form, like the headline
    kernels, roof, enc_util, t_flow, fl, ups = kernel_report(wl, kernel_reps)
    fps = wl.B * steps / wall
    rec = {"batch": wl.B, "value": round(fps, 1), "ms_per_step": round(wall / steps * 1e3, 4),
           "tflops": round(fl["forward_total"] * steps / wall / 1e12, 2), "encoder_mfma_util": enc_util,
           "dominant": roof["kernel"].split("<")[0], "dominant_frac": roof["frac"], "upsample_frac": ups["frac"],
           "tail_frac": ups["tail_frac"]}
    if oracle_sample:
        p = parity_vs_oracle(wl, oracle_sample)
        rec["parity_max_abs_over_ymax"] = p["max_abs"] / p["ymax"]
        rec["parity_rel_dMSE"] = p["rel_dMSE"]
    if c["model_dim"] in (128, 256):      # the opt-in split-precision tier on the same workload (reported separately)
        from adafortitran_amd import _abi
        wl.cfg.precision = _abi.AFT_PRECISION_BF16X3
        wl.eng.invalidate_packed()
        w2, _, _ = timed_steps(wl, wl.step, max(steps // 2, 5), 2, torch.cuda.synchronize)
        rec["split_precision_value"] = round(wl.B * max(steps // 2, 5) / w2, 1)
        if oracle_sample:
            rec["split_precision_max_abs_over_ymax"] = (lambda q: q["max_abs"] / q["ymax"])(parity_vs_oracle(wl, oracle_sample))
    del wl
    torch.cuda.empty_cache()
    return rec


class unsplit:
    """AFT_LANES=1 for the block: the library then runs every forward as ONE launch sequence (include/adafortitran_amd.h "Lanes") --
    what the per-kernel accounting (aft_profile_kernel_f32 replays a kernel class on the UNSPLIT workspace layout) is defined on.  A
    library switch (aft_set_switch: the library never reads the environment on a call path).  At the headline batch the library does
    not split anyway (2.9 rounds of the persistent grids)."""

    def __enter__(self):
        from adafortitran_amd import _lib
        self.old = _lib.get_switch("AFT_LANES")
        _lib.set_switch("AFT_LANES", 1)

    def __exit__(self, *exc):
        from adafortitran_amd import _lib
        _lib.set_switch("AFT_LANES", self.old)


def lanes_of(eng, batch):
    import ctypes as C
    lanes, frames, offs = C.c_int(), (C.c_int * 4)(), (C.c_size_t * 4)()
    eng.lib.aft_workspace_lanes(C.byref(eng.cfg), batch, C.byref(lanes), frames, offs)
    return lanes.value


def batch_sweep(c, device, batches, steps=20, warmup=5):
    """Off-design batches (VERDICT r4 missing #4: the reference's default batch is 64, parser.py:81, and its evaluator runs whatever
    --batch_size says): frames/s of forward + MSE partial and the dominant kernel's fraction of the fp32 roof at each batch, bounded
    to a few seconds.  `rel` = per-frame rate relative to the config's stated batch."""
    import torch
    from adafortitran_amd.hip_ops import profile_kernel
    out, base = {}, None
    L = c["num_layers"]
    for B in batches:
        wl = Workload(c, device, batch=B)
        for _ in range(30):
            wl.forward()            # sustained clocks + a filled workspace for the replay below
        wall, _, _ = timed_steps(wl, wl.step, steps, warmup, torch.cuda.synchronize)
        fl = algorithmic_flops(c, B)
        rec = {"value": round(B * steps / wall, 1), "ms_per_step": round(wall / steps * 1e3, 4), "lanes": lanes_of(wl.eng, B)}
        dom_ms = {}
        with unsplit():             # the dominant kernel's fraction: that kernel class over the WHOLE batch, launched alone
            for _ in range(3):
                wl.forward()        # the workspace in the unsplit layout for the replay
            for which in ("chain", "attention"):
                reps = 10 if c["model_dim"] <= 128 else 3
                profile_kernel(wl.eng, which, B, 2, None)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                profile_kernel(wl.eng, which, B, reps, None)
                e1.record()
                e1.synchronize()
                dom_ms[which] = e0.elapsed_time(e1) / reps
        dom = max(("chain", "attention"), key=lambda k: dom_ms[k] * ((L - 1) if k == "chain" else L))
        rec["dominant"] = dom
        rec["dominant_frac"] = round(fl[dom] / dom_ms[dom] / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4)
        out[str(B)] = rec
        if B == c["batch"]:
            base = rec["value"]
        del wl
        torch.cuda.empty_cache()
    # compact (the driver parses the line): parallel arrays; rel = per-frame rate relative to the config's stated batch
    keys = list(out)
    rec = {"batch": [int(k) for k in keys], "value": [out[k]["value"] for k in keys], "ms_per_step": [out[k]["ms_per_step"] for k in keys],
           "dominant_frac": [out[k]["dominant_frac"] for k in keys], "dominant": sorted({out[k]["dominant"] for k in keys}),
           "lanes": [out[k]["lanes"] for k in keys]}
    if base:
        rec["rel"] = [round(out[k]["value"] / base, 4) for k in keys]
    return rec


def general_shape_rate(c, device, steps=5, warmup=2):
    """A configuration the packed engine does not take (aft_engine_of = GENERAL): frames/s of forward + MSE partial at the config's
    batch and the whole forward's algorithmic FLOP rate as a fraction of the fp32 roof (no per-kernel replay: aft_profile_kernel_f32
    knows the packed engine's classes only)."""
    import torch
    wl = Workload(c, device)
    wall, _, _ = timed_steps(wl, wl.step, steps, warmup, torch.cuda.synchronize)
    fl = algorithmic_flops(c, wl.B)
    rec = [round(wl.B * steps / wall, 1), round(fl["forward_total"] * steps / wall / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)]
    del wl
    torch.cuda.empty_cache()
    return rec


def split_precision_record(c, device, steps, warmup, oracle_sample):
    """The opt-in split-precision tier (aft_config.precision = AFT_PRECISION_BF16X3: encoder GEMMs and attention products on
    bf16 hi/lo terms, fp32 accumulation) on the headline workload -- REPORTED SEPARATELY, never `value`."""
    import torch
    from adafortitran_amd import _abi
    wl = Workload(c, device)
    wl.cfg.precision = _abi.AFT_PRECISION_BF16X3
    wall, dev_ms, per = timed_steps(wl, wl.step, steps, warmup, torch.cuda.synchronize)
    p = parity_vs_oracle(wl, oracle_sample)
    rec = {"dtype": "bf16x3 split operands, f32 accumulate", "value": round(wl.B * steps / wall, 1),
           "ms_per_step": round(wall / steps * 1e3, 4), "max_abs_over_ymax": p["max_abs"] / p["ymax"], "rel_dMSE": p["rel_dMSE"],
           "tol_max_abs_over_ymax": 1e-3, "tol_rel_dMSE": 1e-2}
    del wl
    torch.cuda.empty_cache()
    return rec


def linear_record(device, steps, warmup):
    """BASELINE config 1: LinearEstimator, 120x14 grid, batch 32 -- plumbing only.  Module surface (CPU complex64 pilots in),
    plane-wise; checked against the oracle's linear restatement."""
    import numpy as np
    import torch
    import adafortitran_amd as A
    from adafortitran_amd import synth
    from oracle import oracle
    B = 32
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    mc = A.ModelConfig(model_type="linear", patch_size=(3, 2), num_layers=1, model_dim=32, num_head=1, device="cuda")
    model = A.LinearEstimator(sc, mc).eval()
    w = synth.uniform_pm(SEED, "linear.weight", (1680, 24), 1 / np.sqrt(24))
    b = synth.uniform_pm(SEED, "linear.bias", (1680,), 1 / np.sqrt(24))
    model.load_state_dict({"linear.weight": torch.from_numpy(w), "linear.bias": torch.from_numpy(b)})
    inp = synth.make_inputs(B, seed=SEED)
    pil_cpu = torch.from_numpy(inp["pilots"])

    def step():
        with torch.no_grad():
            model(pil_cpu)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    with torch.no_grad():
        got = model(pil_cpu).cpu().numpy()
    ref = oracle.linear_forward(w, b, inp["pilots"], (120, 14))
    return {"batch": B, "value": round(B * steps / wall, 1), "ms_per_step": round(wall / steps * 1e3, 4),
            "parity_max_abs": float(np.abs(got - ref).max())}


def next_rows(wl):
    """SURVEY 8(f) rows on this GPU, bounded to a few seconds: f1 training step (forward + backward + fused Adam, B = 128,
    dropout 0.1) with its fraction of the fp32-MFMA roof (3 x forward FLOPs per step), f2 pilot gather, f3 evaluation
    sweep over a PackedLoader, f4 LS baseline (tools/train_bench.py, tools/next_rows_bench.py hold the measurement code)."""
    import importlib.util
    out = {}
    for name in ("train_bench", "next_rows_bench"):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out[name] = mod
    fl = algorithmic_flops(C3, 128)
    tr = out["train_bench"].measure(batch=128, steps=10, warmup=3, model_name="adafortitran", dropout=0.1, modes=("hip",))
    rec = {"f1_train_step_ms": round(tr["hip"], 3),
           "f1_frac_of_fp32_roof": round(3 * fl["forward_total"] / (tr["hip"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
    # the reference's default batch (src/main/parser.py:81): 64 frames per step
    tr64 = out["train_bench"].measure(batch=64, steps=10, warmup=3, model_name="adafortitran", dropout=0.1, modes=("hip",))
    rec["f1_train_step_ms_batch64"] = round(tr64["hip"], 3)
    nr = out["next_rows_bench"].measure(frames=4096, batch=128)
    rec.update({"f2_gather_GBs": nr["f2_pilot_gather"]["GB_per_s"], "f2_gather_hbm_frac": nr["f2_pilot_gather"]["frac_of_hbm_peak"],
                "f2_loader_fps": nr["f2_packed_loader"]["frames_per_s"],
                "f3_sweep_fps": nr["f3_eval_sweep"]["frames_per_s_device_accumulator"],
                "f3_item_loop_fps": nr["f3_eval_sweep"]["frames_per_s_item_per_batch"],
                "f4_ls_GBs": nr["f4_ls_mse_db"]["GB_per_s"], "f4_ls_hbm_frac": nr["f4_ls_mse_db"]["frac_of_hbm_peak"]})
    return rec


def cpu_baseline(wl):
    """Reference-equivalent CPU path: this package's estimator on device='cpu' is assembled from the same torch.nn
    modules as the reference (nn.Linear / Conv2d / TransformerEncoder fast path); pinned against the imported reference
    in the build container (tests/golden/cpu_standin.json: outputs bit-identical, forward time ratio 0.995-0.998).
    Bounded: a thread sweep + <= 10 forwards of the same B=128 batch."""
    import numpy as np
    import torch
    from adafortitran_amd import synth
    model = make_module(wl.c, "cpu", wl.sd)
    pil = torch.from_numpy(wl.inp["pilots"])
    meta = synth.meta_tuple(wl.inp) if wl.adaptive else None
    call = (lambda: model(pil, meta)) if wl.adaptive else (lambda: model(pil))
    info = host_cpu_info()
    default_threads = torch.get_num_threads()
    # The baseline moved 13 % between rounds (143.9 frames/s at 16 threads, 124.5 at 32: the sweep picked on ONE forward per
    # candidate and the threads roamed over both sockets).  Now: the process is pinned to ONE socket's physical cores (one logical
    # CPU per core) for the baseline, the candidates are thread counts up to that many cores, each candidate is the MEDIAN of three
    # forwards behind an untimed one, and the table goes into the record.
    # sched_setaffinity(0, ..) moves the CALLING thread only, and torch's intra-op pool exists by now: every thread of the
    # process (/proc/self/task) is pinned, threads created later inherit the mask, and the mask actually in force goes into the record.
    cores = one_socket_cores()
    old_affinity, pinned_threads = None, 0

    def pin_all(cpus):
        n = 0
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cpus)
                n += 1
            except (OSError, ValueError):
                pass
        return n

    try:
        if cores and hasattr(os, "sched_setaffinity"):
            old_affinity = os.sched_getaffinity(0)
            usable = sorted(set(cores) & old_affinity)
            if usable:
                pinned_threads = pin_all(usable)
                cores = usable
    except OSError:
        old_affinity = None
    ncore = len(cores) if cores else max(1, info["physical_cores"] // max(1, info["sockets"]))
    sweep = {}
    with torch.no_grad():
        t_budget = time.perf_counter() + 24.0
        for cand in sorted({c for c in (4, 8, 16, 32, 64, ncore) if c <= ncore}):
            torch.set_num_threads(cand)
            call()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                call()
                ts.append(time.perf_counter() - t0)
            sweep[cand] = float(np.median(ts))
            if time.perf_counter() > t_budget:
                break
        best_t = min(sweep, key=sweep.get)
        torch.set_num_threads(best_t)
        times = []
        t_end = time.perf_counter() + 8.0
        while len(times) < 5 or (time.perf_counter() < t_end and len(times) < 9):
            t0 = time.perf_counter()
            call()
            times.append(time.perf_counter() - t0)
        torch.set_num_threads(default_threads)
    affinity_used = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    if old_affinity is not None:
        try:
            pin_all(old_affinity)
        except OSError:
            pass
    med = float(np.median(times))
    out = {"value": round(wl.B / med, 2), "unit": "frames/s", "cores": best_t, "kind": "port",
           "sample": f"{len(times)} forwards of B={wl.B} (same workload), median {med * 1e3:.0f} ms; threads chosen by a sweep (median of 3 "
                     f"forwards each) on one socket's {ncore} physical cores; torch {torch.__version__} CPU composite",
           "thread_sweep_fps": {str(k): round(wl.B / v, 1) for k, v in sorted(sweep.items())},
           "spread": round((max(times) - min(times)) / med, 3), "affinity_cpus": affinity_used, "threads_pinned": pinned_threads,
           "cpu_model": info["cpu_model"], "sockets": info["sockets"], "physical_cores": info["physical_cores"]}
    try:    # the C oracle (oracle/aft_oracle.c) on a smaller bounded sample, for the record
        from oracle import oracle
        orc = oracle.Oracle(wl.cfg, wl.sd)
        nb = 8
        args = [wl.inp[k][:nb] for k in ("snr", "ds", "dop")] if wl.adaptive else [None] * 3
        t0 = time.perf_counter()
        orc.forward(wl.inp["pilots"][:nb], *args)
        out["oracle_c_fps"] = round(nb / (time.perf_counter() - t0), 2)
        out["oracle_c_threads"] = oracle.num_threads()
    except Exception as exc:  # the oracle is optional test infrastructure
        out["oracle_c_error"] = str(exc)[:80]
    return out


def load_pmc_traffic():
    """HBM bytes per chain-kernel launch from the committed rocprofv3 --pmc summary (a constant of the committed
    profile -- profiles/pmc_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE -- not something this run observed)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            return json.load(fh).get("chain_bytes_per_launch")
    except Exception:
        return None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="per-rank workload (default C3)")
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU per step (default: the config's)")
    ap.add_argument("--model", default=None, choices=["adafortitran", "fortitran"], help="alias: adafortitran = C3, fortitran = C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the N=1 legs (module surface, configs, parity, next rows, CPU)")
    ap.add_argument("--verbose-json", default="", help="also write the annotated record to this path")
    ap.add_argument("--stub", action="store_true", help="tests only: control flow on CPU/gloo, measures nothing")
    ap.add_argument("--share-gpu", action="store_true", help="tests only: every rank on device 0, collectives over gloo (RCCL refuses two "
                    "ranks on one device); runs the N-rank path with the real kernels on a 1-GPU box, the numbers mean nothing")
    args = ap.parse_args()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args, sys.argv[1:])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2

    import numpy as np
    import torch
    dist = None
    if args.stub:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            print("bench.py needs an MI355X: the HIP path has no CPU fallback", file=sys.stderr)
            return 2
        if args.share_gpu:
            local_rank = 0
        if torch.cuda.device_count() <= local_rank:
            print(f"bench.py: rank {rank} has no device {local_rank}", file=sys.stderr)
            return 2
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    if "WORLD_SIZE" in os.environ:   # under torchrun the RCCL path runs for every world size, 1 included
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.stub or args.share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)  # "nccl" is RCCL on ROCm

    head = dict(CONFIGS[args.config or ("C2" if args.model == "fortitran" else "C3")])
    wl = (StubWorkload if args.stub else Workload)(head, device, rank=rank, batch=args.batch or None)   # a different shard of frames per rank
    B = wl.B

    def fence():
        wl.sync()
        if dist is not None:
            dist.barrier()
            wl.sync()

    # Device spin-up, part of the setup (untimed, before the W warm-up steps): the MI355X reaches its sustained clocks and the
    # caches their steady state only after a few hundred ms of work -- with the driver's `--warmup 5` (8 ms) the first timed steps
    # ran 5 % slow in round 2 (p10 / p90 = 1.60 / 1.77 ms against 1.57 / 1.59 ms in a 200-step run).  The metric is steady-state
    # frames/s, so the bench warms the device for PREWARM steps itself and says so in the line.
    # The timed step is the metric's own form: the module surface fed CPU tensors, H2D inside the step.
    for _ in range(0 if args.stub else PREWARM_STEPS):
        wl.surface_step()
    wl.sync()
    for _ in range(args.warmup):
        wl.surface_step()
    wl.acc.sum_sq.zero_()
    wl.acc.n_elements = 0
    elapsed_local, dev_ms, per_step = timed_steps(wl, wl.surface_step, args.steps, 0, fence)
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]   # noqa: E731
    elapsed, per_rank = elapsed_local, None
    if dist is not None:
        # MAX over ranks of the barrier-to-barrier wall time is the step time of the job; the per-rank DEVICE time of the
        # same K steps (no barrier wait inside) says which rank, if any, was slow
        t = torch.tensor([elapsed_local, dev_ms / 1e3], dtype=torch.float64, device=device)
        all_t = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(all_t, t)
        walls, devs = [float(x[0]) for x in all_t], [float(x[1]) for x in all_t]
        elapsed = max(walls)
        per_rank = {"device_ms_per_step_min": round(min(devs) / args.steps * 1e3, 4),
                    "device_ms_per_step_max": round(max(devs) / args.steps * 1e3, 4),
                    "slowest_rank": int(np.argmax(devs))}

    if dist is not None:
        # who took part (so that a SCALE record proves N distinct GPUs by itself): every rank's (RANK, LOCAL_RANK, device
        # index, PCI bus id, device name, host, pid) through the process group; two ranks on one PCI device => exit 4
        # (--share-gpu and --stub are test modes and say so in `data`)
        ident = rank_identity(rank, local_rank, device, args)
        ranks = [None] * world
        dist.all_gather_object(ranks, ident)
        per_rank["world_size"] = dist.get_world_size()
        per_rank["backend"] = dist.get_backend()
        per_rank["collective_lib"] = collective_lib_version(args)
        per_rank["ranks"] = [[r["rank"], r["local_rank"], r["device"], r["pci"]] for r in ranks]   # compact: the line is parsed
        per_rank["rank_fields"] = "rank,local_rank,device,pci"
        per_rank["device_names"] = sorted({r.get("name", "cpu") for r in ranks})
        per_rank["hosts"] = sorted({r["host"] for r in ranks})
        per_rank["pids"] = len({(r["host"], r["pid"]) for r in ranks})
        # a device is (host, PCI id): equal PCI ids on different hosts are different GPUs; without a PCI id or uuid from torch the
        # (host, device index) fallback only says what LOCAL_RANK selected, so the check then warns instead of failing
        bus = [(r["host"], r["pci"]) for r in ranks]
        per_rank["distinct_devices"] = len(set(bus))
        known = all(r.get("pci_known", True) for r in ranks)
        if not (args.stub or args.share_gpu) and len(set(bus)) != world and not known:
            print(f"bench.py: warning: device identities unavailable from torch (index fallback): {bus}", file=sys.stderr)
        if not (args.stub or args.share_gpu) and len(set(bus)) != world and known:
            print(f"bench.py: {world} ranks but only {len(set(bus))} distinct (host, PCI) devices: {bus}", file=sys.stderr)
            dist.barrier()
            dist.destroy_process_group()
            return 4

    # ---- end-of-sweep metric: one all-gather of (sum|e|^2, n_frames) per rank ----
    fence()
    t0 = time.perf_counter()
    mse = wl.acc.result()   # RCCL all-gather of the 16-byte (sum, n) pairs when a process group exists
    t_first = time.perf_counter() - t0
    if dist is not None:    # the collective's own latency once everything is warm (the first one may set up the ring)
        reps = 10
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            wl.acc.result()
        per_rank["metric_allgather_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        per_rank["metric_allgather_first_ms"] = round(t_first * 1e3, 4)

    if rank == 0:
        frames = B * world * args.steps
        fps = frames / elapsed
        result = {
            "metric": "OFDM frames/sec (120x14 grid, batch 128) + channel-estimation MSE vs reference",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "prewarm_steps": 0 if args.stub else PREWARM_STEPS,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "stub" if args.stub else ("synthetic (TEST RUN: all ranks share device 0, gloo)" if args.share_gpu else "synthetic"),
            "config": {"workload": f"{head['name']}: {head['label']}; {B} frames per GPU per step; step = model(pilots_cpu, meta_cpu) through the "
                                   "module surface + device MSE partial; value = SURVEY 8(d)'s metric: H2D of pilots + meta INSIDE the step, output left on "
                                   "the device (= value_h2d_inclusive); the bare engine call on inputs resident in HBM is value_resident; stateless "
                                   "aft_forward_f32 (encoder weights re-laid inside every call)",
                       "frames_per_gpu": B, "global_batch": B * world, "parallelism": f"frames sharded over {world} rank(s)"},
            "device_ms_per_step": round(dev_ms / args.steps, 4),
            "device_step_ms": {"p10": round(pct(0.10), 4), "p50": round(pct(0.50), 4), "p90": round(pct(0.90), 4)},
            "mse_db_vs_random_target": round(10 * np.log10(mse), 4),
        }
        if per_rank is not None:
            result["per_rank"] = per_rank
        if not args.stub:
            result["config"]["lanes"] = lanes_of(wl.eng, B)    # launch sequences the timed forward ran as (1 at the stated batches)
            kernels, roof, enc_util, t_flow, fl, ups = kernel_report(wl, 20)
            roof["traffic"] = load_pmc_traffic()
            result.update({"roofline": roof, "kernels": kernels, "upsampler": ups, "forward_ms_one_event_pair": t_flow,
                           "encoder_mfma_util": enc_util,
                           "whole_path_tflops": round(fl["forward_total"] * world * args.steps / elapsed / 1e12, 2)})
        if world == 1 and not args.headline_only and not args.stub:
            result["parity"] = parity_vs_oracle(wl, 8 if head["name"] != "C5" else 1)
            result["value_h2d_inclusive"] = result["value"]
            ms = module_surface(wl, args.steps, args.warmup, fps)
            result["value_resident"] = ms["resident_value"]
            result["module_surface"] = ms
            cfgs = {"C1": linear_record(device, 200, 20)}
            for other in (C2, C3, C5):
                if other["name"] == head["name"]:
                    cfgs[other["name"]] = {"batch": B, "value": result["value"], "ms_per_step": result["ms_per_step"],
                                           "encoder_mfma_util": enc_util, "dominant_frac": roof["frac"],
                                           "upsample_frac": ups["frac"], "tail_frac": ups["tail_frac"],
                                           "parity_rel_dMSE": result["parity"]["rel_dMSE"]}
                elif other["name"] == "C5":
                    cfgs["C5"] = config_record(C5, device, 20, 3, 1, 3)
                else:
                    cfgs[other["name"]] = config_record(other, device, 100, 10, 8, 10)
            result["configs"] = cfgs     # C4 = C3 with --gpus 8; C5 as an 8-GPU config = --config C5 --gpus 8
            try:
                result["batch_sweep"] = {head["name"]: batch_sweep(head, device, (16, 32, 64, 96, 128, 129, 192, 256, 512)
                                                                   if head["name"] != "C5" else (16, 32, 64, 65))}
                if head["name"] != "C5":
                    result["batch_sweep"]["C5"] = batch_sweep(C5, device, (16, 32, 64, 65), steps=5, warmup=2)
            except Exception as exc:
                result["batch_sweep"] = {"error": f"{type(exc).__name__}: {exc}"[:160]}
            # the shapes DESIGN.md calls "covered, not tuned" (VERDICT r4 weak #6 / item 7): other model dims and head dims on the default
            # grid, and a grid the 120 x 14 conv kernels do not serve -- rate and dominant-kernel fraction at B = 128, a second each
            try:
                others = {}
                for tag, over in (("d64_h2", dict(model_dim=64, num_head=2)), ("d96_h3", dict(model_dim=96, num_head=3)),
                                  ("d192_h6", dict(model_dim=192, num_head=6)), ("d256_h8", dict(model_dim=256, num_head=8)),
                                  ("d128_h8_hd16", dict(num_head=8)), ("d128_h2_hd64", dict(num_head=2)),
                                  ("d128_h16_hd8", dict(num_head=16)), ("d96_h4_hd24", dict(model_dim=96, num_head=4)),
                                  ("d192_h4_hd48", dict(model_dim=192, num_head=4)),
                                  ("grid_60x14", dict(ofdm=(60, 14), pilot=(6, 2), hidden=(7, 42, 280))),
                                  ("grid_12x14_28tok", dict(ofdm=(12, 14), pilot=(4, 2), hidden=(7, 42, 56)))):
                    r = batch_sweep(dict(C3, **over, batch=128), device, (128,), steps=10, warmup=3)
                    others[tag] = [r["value"][0], r["dominant_frac"][0]]
                    # the GENERAL engine's shapes (round 6, VERDICT r5 item 3: covered, not tuned -- row-major GEMM / attention / LayerNorm launches):
                # frames/s and the whole forward's fraction of the fp32 roof (algorithmic FLOPs of the config / step time)
                for tag, over in (("d512_h8", dict(model_dim=512, num_head=8)), ("d512_h4_hd128", dict(model_dim=512, num_head=4)),
                                  ("d256_h2_hd128", dict(model_dim=256, num_head=2)), ("d224_h4_hd56", dict(model_dim=224, num_head=4)),
                                  ("d200_h8_hd25", dict(model_dim=200, num_head=8))):
                    others["general_" + tag] = general_shape_rate(dict(C3, **over, batch=128), device)
                result["other_shapes"] = {"fields": "frames/s, dominant kernel's fraction of the fp32 roof (B=128, adaptive); general_*: "
                                                    "frames/s, whole forward's fraction of the fp32 roof", **others}
            except Exception as exc:
                result["other_shapes"] = {"error": f"{type(exc).__name__}: {exc}"[:160]}
            if head["model_dim"] in (128, 256):
                result["split_precision"] = split_precision_record(head, device, min(args.steps, 100), min(args.warmup, 10), 8)
            try:
                result["next_rows"] = next_rows(wl)
            except Exception as exc:     # never lose the headline to a secondary leg
                result["next_rows"] = {"error": f"{type(exc).__name__}: {exc}"[:160]}
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(wl)
                result["speedup_vs_cpu"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
        if args.verbose_json:
            with open(args.verbose_json, "w") as fh:
                json.dump(result, fh, indent=1)
        print(json.dumps(result, separators=(",", ":")))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
