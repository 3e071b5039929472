#!/usr/bin/env python3
"""Training step time (forward + loss + backward + Adam) at the bench workload, B frames per GPU:
the HIP training kernels vs the same module differentiated by PyTorch-ROCm only.

    python tools/train_bench.py [--gpus N] [--batch 128] [--steps 20] [--model adafortitran] [--only hip]

``--gpus N`` (N > 1, no torchrun environment): this process touches no GPU and starts
``torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1`` on itself, as bench.py does.  Under torchrun every
rank trains on its own shard of frames (B per GPU, weak scaling) and ``ShardedFlatAdam.step()`` runs its
reduce-scatter (gradients) -> fused Adam on the rank's shard -> all-gather (parameters) over RCCL INSIDE the timed
step (reference caller: ``TrainingLoop.train_epoch``, src/main/trainer.py:195-233).  Rank 0 prints one JSON line with
the step time (MAX over ranks of the barrier-to-barrier wall time), the per-rank device times (min / max) and the
cost of the two collectives alone on the flat 3.95 MB buffers."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(name, dropout, device="cuda"):
    import adafortitran_amd as A
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    kw = dict(model_type=name, patch_size=(3, 2), num_layers=6, model_dim=128, num_head=4, activation="gelu",
              max_seq_len=512, pos_encoding_type="learnable", device=device, dropout=dropout)
    if name == "adafortitran":
        kw.update(channel_adaptivity_hidden_sizes=[7, 42, 560], adaptive_token_length=6)
    cls = A.AdaFortiTranEstimator if name == "adafortitran" else A.FortiTranEstimator
    return cls(sc, A.ModelConfig(**kw))


def measure(batch=128, steps=20, warmup=5, model_name="adafortitran", dropout=0.1, modes=("hip", "torch"), dense="hip",
            optimizer="flat", dist=None, rank=0, detail=None):
    """ms per training step for each mode; with a process group the steps are bracketed by barriers and the value is the
    MAX over ranks.  ``detail`` (dict) receives per-rank device times and the collectives' own cost."""
    import torch
    from adafortitran_amd import synth, training
    torch.manual_seed(0)
    model = build(model_name, dropout).train()
    if optimizer == "flat":
        from adafortitran_amd.optim import ShardedFlatAdam
        opt = ShardedFlatAdam(model.parameters(), lr=1e-3)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    inp = synth.make_inputs(batch, seed=1 + 1000 * rank)          # a different shard of frames per rank
    dev = torch.device("cuda", torch.cuda.current_device())
    pil, tgt = torch.from_numpy(inp["pilots"]).to(dev), torch.from_numpy(inp["target"]).to(dev)
    meta = synth.meta_tuple(inp) if model_name == "adafortitran" else None

    def step():
        opt.zero_grad()
        out = model(pil, meta) if meta is not None else model(pil)
        loss = torch.nn.functional.mse_loss(torch.view_as_real(out), torch.view_as_real(tgt))
        loss.backward()
        opt.step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    res = {}
    for mode in modes:
        model.transformer_encoder.hip_training = mode == "hip"
        model.initial_enhancer.hip_training = model.final_refiner.hip_training = mode == "hip"
        training.HipLinear.default_hip_training = mode == "hip" and dense == "hip"
        if hasattr(model, "channel_adapter"):
            model.channel_adapter.hip_training = mode == "hip"
        for _ in range(warmup):
            step()
        fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        fence()
        wall = time.perf_counter() - t0
        dev_s = e0.elapsed_time(e1) / 1e3
        if dist is not None:
            t = torch.tensor([wall, dev_s], dtype=torch.float64, device=dev)
            all_t = [torch.empty_like(t) for _ in range(dist.get_world_size())]
            dist.all_gather(all_t, t)
            wall = max(float(x[0]) for x in all_t)
            if detail is not None:
                devs = [float(x[1]) / steps * 1e3 for x in all_t]
                detail[mode] = {"device_ms_per_step_min": round(min(devs), 3), "device_ms_per_step_max": round(max(devs), 3)}
        res[mode] = wall / steps * 1e3
    if dist is not None and optimizer == "flat" and detail is not None:
        # the two collectives of ShardedFlatAdam.step() alone, on the real flat buffers (3.95 MB each way)
        flat = opt.flat
        shard = torch.empty(flat.padded // dist.get_world_size(), dtype=torch.float32, device=dev)
        for _ in range(3):
            dist.reduce_scatter_tensor(shard, flat.grad)
            dist.all_gather_into_tensor(flat.data, opt.p_shard.clone())
        fence()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            dist.reduce_scatter_tensor(shard, flat.grad)
            dist.all_gather_into_tensor(flat.data, opt.p_shard.clone())
        fence()
        detail["collectives_ms_per_step"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        detail["flat_buffer_bytes"] = int(flat.padded * 4)
    training.HipLinear.default_hip_training = True
    return res


def self_launch(a, argv):
    import torch
    if not a.share_gpu and torch.cuda.device_count() < a.gpus:
        print(f"train_bench.py: --gpus {a.gpus} but only {torch.cuda.device_count()} device(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="adafortitran")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--only", default="", help="hip | torch")
    ap.add_argument("--dense", default="hip", choices=["blas", "hip"], help="thin dense layers: the library GEMM (default) or hipBLASLt")
    ap.add_argument("--optimizer", default="flat", choices=["flat", "torch"], help="flat = ShardedFlatAdam (fused kernel)")
    ap.add_argument("--share-gpu", action="store_true", help="tests only: every rank on device 0, collectives over gloo (RCCL refuses "
                    "two ranks on one device); the numbers mean nothing")
    a = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return self_launch(a, sys.argv[1:])
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if world != a.gpus:
        print(f"train_bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    import torch
    if a.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if "WORLD_SIZE" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    modes = (a.only,) if a.only else (("hip", "torch") if world == 1 else ("hip",))
    detail = {}
    res = measure(a.batch, a.steps, a.warmup, a.model, a.dropout, modes, a.dense, a.optimizer, dist, rank, detail)
    if rank == 0:
        out = {"metric": "training step (fwd+bwd+Adam) ms", "n_gpus": world, "batch_per_gpu": a.batch, "model": a.model,
               "dropout": a.dropout, "ms_per_step": {k: round(v, 3) for k, v in res.items()},
               "frames_per_s": {k: round(a.batch * world / v * 1e3, 1) for k, v in res.items()}}
        if len(res) == 2:
            out["speedup_vs_pytorch_rocm"] = round(res["torch"] / res["hip"], 3)
        if detail:
            out["per_rank"] = detail
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
