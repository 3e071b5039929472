#!/usr/bin/env python3
"""bench.py -- OFDM frames/s of the AdaFortiTran forward path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model adafortitran|fortitran] [--batch 128]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic input on every rank: one
``aft_forward_f32`` (B=128 complex 120x14 frames, inputs already resident in HBM) plus the
device-side channel-MSE partial sum.  One process per GPU; frames shard across ranks with no
data-path collective (weak scaling: 128 frames per GPU); a single RCCL all-gather of the
per-rank (sum|e|^2, n) pair closes the sweep.  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline      -- dominant kernel (chain: out-proj+LN1+FFN+LN2+QKV) vs the fp32-MFMA roof,
                   duration measured live with events on the launch stream;
  kernels       -- the same measurement for every kernel class of the forward;
  cpu_baseline  -- the reference-equivalent CPU path (same torch.nn modules => same ATen /
                   oneDNN / MKL kernels as the reference) timed on this box's host cores on a
                   bounded sample, rank 0, N=1 only; the C oracle's rate rides along.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SPEC = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4)
HIDDEN = (7, 42, 560)
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = fp32 vector peak
PEAK_HBM_GBS = 8000.0


def algorithmic_flops(spec, tokens, planes):
    """FLOPs per launch (MAC = 2) of each kernel class -- SURVEY.md 2.2 / 8(d) per-plane figures
    x the planes one launch processes (DESIGN.md 'roofline accounting')."""
    d, L = spec["model_dim"], spec["num_layers"]
    rows = planes * tokens
    qkv = 2 * rows * d * 3 * d
    proj = 2 * rows * d * d
    ffn = 2 * rows * d * 2 * d * 2
    attn = 2 * 2 * planes * tokens * tokens * d      # QK^T + PV over all heads
    conv = planes * 15_966_720                        # 4 convs, SURVEY.md 2.2 K2 (120x14 grid)
    up = planes * 80_640
    return {"qkv": qkv, "chain": proj + ffn + qkv, "chain_last": proj + ffn, "attention": attn, "upsample": conv + up,
            "tail": conv + 2 * planes * tokens * d * 6, "embed": 2 * rows * d * 12,
            "encoder_total": L * (qkv + proj + ffn + attn)}


def time_kernels_in_flow(eng, batch, reps, pil, out, layers):
    """Average duration (ms) of every kernel class, launched in the order of a real forward (conv head,
    embed, QKV, [attention, chain] x (L-1), attention, last chain, conv tail) with one event pair around
    each launch, recorded on the stream the library launches on (torch's current stream).  Measuring
    the dominant kernel between its real neighbours keeps it at the clocks and cache state it has inside
    the timed region (a loop of the same kernel alone runs 6-8 % slower: sustained fp32-MFMA power)."""
    from adafortitran_amd.hip_ops import profile_kernel
    flow = [("upsample", pil), ("embed", None), ("qkv", None)]
    for _ in range(layers - 1):
        flow += [("attention", None), ("chain", None)]
    flow += [("attention", None), ("chain_last", None), ("tail", out)]
    for which, io in flow:
        profile_kernel(eng, which, batch, 1, io)
    torch.cuda.synchronize()
    pairs = []
    for _ in range(reps):
        for which, io in flow:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            profile_kernel(eng, which, batch, 1, io)
            e1.record()
            pairs.append((which, e0, e1))
    torch.cuda.synchronize()
    tot, cnt = {}, {}
    for which, e0, e1 in pairs:
        tot[which] = tot.get(which, 0.0) + e0.elapsed_time(e1)
        cnt[which] = cnt.get(which, 0) + 1
    return {k: tot[k] / cnt[k] for k in tot}


def cpu_baseline(model_name, batch, sd, inp):
    """Reference-equivalent CPU path: this package's estimator on device='cpu' is assembled from
    the same torch.nn modules as the reference (nn.Linear / Conv2d / TransformerEncoder fast path),
    validated against reference outputs in tests/test_estimators_cpu.py."""
    import adafortitran_amd as A
    from adafortitran_amd import synth
    adaptive = model_name == "adafortitran"
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    kw = dict(model_type=model_name, patch_size=(3, 2), num_layers=6, model_dim=128, num_head=4, device="cpu")
    if adaptive:
        kw.update(channel_adaptivity_hidden_sizes=list(HIDDEN), adaptive_token_length=6)
    model = (A.AdaFortiTranEstimator if adaptive else A.FortiTranEstimator)(sc, A.ModelConfig(**kw)).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    pil = torch.from_numpy(inp["pilots"])
    meta = synth.meta_tuple(inp) if adaptive else None
    call = (lambda: model(pil, meta)) if adaptive else (lambda: model(pil))
    host_cores = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    with torch.no_grad():
        # intra-op thread count that serves the reference best on this host (oversubscribing a
        # 280-token problem with every hardware thread is slower than a moderate count)
        best_t, best_dt = default_threads, float("inf")
        for cand in sorted({8, 16, 32, 64, max(1, host_cores // 2), host_cores, default_threads}):
            if cand > host_cores:
                continue
            torch.set_num_threads(cand)
            call()
            t0 = time.perf_counter()
            call()
            dt = time.perf_counter() - t0
            if dt < best_dt:
                best_t, best_dt = cand, dt
        torch.set_num_threads(best_t)
        threads = best_t
        times = []
        t_end = time.perf_counter() + 10.0
        while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 10):
            t0 = time.perf_counter()
            call()
            times.append(time.perf_counter() - t0)
        torch.set_num_threads(default_threads)
    med = float(np.median(times))
    out = {"value": batch / med, "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"{len(times)} forwards of B={batch} (same workload), median {med * 1e3:.0f} ms, best of a "
                     f"thread sweep on {host_cores} logical CPUs; torch {torch.__version__} nn-module composite = "
                     f"the reference's own ATen/oneDNN/MKL CPU kernels, eval()+no_grad(), fp32"}
    # the C oracle (oracle/aft_oracle.c) on a smaller bounded sample, for the record
    try:
        from adafortitran_amd import _abi
        from oracle import oracle
        cfg = _abi.make_config(**SPEC, adaptive_hidden=HIDDEN if adaptive else None)
        orc = oracle.Oracle(cfg, sd)
        nb = 8
        args = [inp[k][:nb] for k in ("snr", "ds", "dop")] if adaptive else [None] * 3
        t0 = time.perf_counter()
        orc.forward(inp["pilots"][:nb], *args)
        dt = time.perf_counter() - t0
        out["oracle_c"] = {"value": nb / dt, "unit": "frames/s", "cores": oracle.num_threads(),
                           "sample": f"1 forward of B={nb}, OpenMP, double accumulation"}
    except Exception as exc:  # the oracle is optional test infrastructure
        out["oracle_c"] = {"error": str(exc)[:120]}
    return out


def load_pmc_traffic():
    """HBM bytes per chain-kernel launch from the committed rocprofv3 --pmc summary, if present."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get("chain_bytes_per_launch")
    except Exception:
        return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128, help="frames per GPU per step")
    ap.add_argument("--model", default="adafortitran", choices=["adafortitran", "fortitran"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)  # "nccl" is RCCL on ROCm

    from adafortitran_amd import _abi, synth
    from adafortitran_amd.hip_ops import engine_from_numpy
    from adafortitran_amd.metrics import MseAccumulator

    adaptive = args.model == "adafortitran"
    B = args.batch
    sd = synth.make_state_dict(**SPEC, adaptive_hidden=HIDDEN if adaptive else None, seed=20251114)
    cfg = _abi.make_config(**SPEC, adaptive_hidden=HIDDEN if adaptive else None)
    eng = engine_from_numpy(cfg, sd, device)
    inp = synth.make_inputs(B, seed=20251114 + 1000 * rank)   # a different shard of frames per rank
    pil = torch.from_numpy(inp["pilots"]).to(device)
    tgt = torch.from_numpy(inp["target"]).to(device)
    meta = [torch.from_numpy(inp[k]).to(device) for k in ("snr", "ds", "dop")] if adaptive else [None] * 3
    out = torch.empty((B, 120, 14), dtype=torch.complex64, device=device)
    acc = MseAccumulator(device)

    def step():
        eng.forward(pil, *meta, out=out)
        acc.update(out, tgt)          # device-side partial sum, no host sync

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    acc = MseAccumulator(device)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    marks = [ev0]                      # one event per step boundary: per-step device times (median, p10/p90)
    for _ in range(args.steps):
        step()
        m = torch.cuda.Event(enable_timing=True)
        m.record()
        marks.append(m)
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    per_step = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]   # noqa: E731
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- end-of-sweep metric: one all-gather of (sum|e|^2, n_frames) per rank ----
    mse = acc.result()   # RCCL all-gather of the 16-byte (sum, n) pairs when world > 1

    if rank == 0:
        frames = B * world * args.steps
        planes, tokens = 2 * B, cfg.tokens
        fl = algorithmic_flops(SPEC, tokens, planes)
        kms = time_kernels_in_flow(eng, B, 20, pil, out, SPEC["num_layers"])
        kernels = {name: {"ms": round(ms, 4), "tflops": round(fl[name] / ms / 1e9, 2)} for name, ms in kms.items()}
        chain_ms = kernels["chain"]["ms"]
        achieved = fl["chain"] / chain_ms / 1e9
        L = SPEC["num_layers"]
        enc_ms = (kernels["qkv"]["ms"] + L * kernels["attention"]["ms"] + (L - 1) * chain_ms
                  + kernels["chain_last"]["ms"])
        result = {
            "metric": "OFDM frames/sec (120x14 grid, batch 128) + channel-estimation MSE vs reference",
            "value": round(frames / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'AdaFortiTran' if adaptive else 'FortiTran'} default config (6 layers, d=128, 4 heads"
                                   f"{', adaptive tokens' if adaptive else ''}), 120x14 grid, pilots 12x2, "
                                   f"batch {B} frames per GPU, forward + device MSE partial, inputs resident in HBM",
                       "frames_per_gpu": B, "global_batch": B * world, "parallelism": f"frames sharded over {world} rank(s)"},
            "device_ms_per_step": round(dev_ms / args.steps, 4),
            "device_step_ms": {"p10": round(pct(0.10), 4), "p50": round(pct(0.50), 4), "p90": round(pct(0.90), 4)},
            "mse_db_vs_random_target": round(10 * np.log10(mse), 4),
            "roofline": {"kernel": "chain_kernel<128,GELU,MLP=true,QKV=true> (out-proj+LN1+FFN+LN2 + next layer's QKV)",
                         "bound": "mfma",
                         "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": load_pmc_traffic(),
                         "flops_per_launch": fl["chain"], "ms_per_launch": chain_ms,
                         "instruction_class": "v_mfma_f32_32x32x2_f32 (exact fp32)"},
            "kernels": kernels,
            "encoder_mfma_util": round(fl["encoder_total"] / enc_ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
            "whole_path_tflops": round((1_429_387_932 if adaptive else 1_428_241_920) * frames / elapsed / 1e12, 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.model, B, sd, inp)
            result["speedup_vs_cpu"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
