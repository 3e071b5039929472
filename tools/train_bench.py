#!/usr/bin/env python3
"""Training step time (forward + loss + backward + Adam) at the bench workload, B frames per GPU:
the encoder on the HIP training kernels vs the same module differentiated by PyTorch-ROCm only.
    python tools/train_bench.py [--batch 128] [--steps 20] [--model adafortitran]
Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import adafortitran_amd as A
from adafortitran_amd import synth, training


def build(name, dropout):
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    kw = dict(model_type=name, patch_size=(3, 2), num_layers=6, model_dim=128, num_head=4, activation="gelu",
              max_seq_len=512, pos_encoding_type="learnable", device="cuda", dropout=dropout)
    if name == "adafortitran":
        kw.update(channel_adaptivity_hidden_sizes=[7, 42, 560], adaptive_token_length=6)
    cls = A.AdaFortiTranEstimator if name == "adafortitran" else A.FortiTranEstimator
    return cls(sc, A.ModelConfig(**kw))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="adafortitran")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--only", default="", help="hip | torch")
    ap.add_argument("--dense", default="hip", choices=["blas", "hip"], help="thin dense layers: the library GEMM (default) or hipBLASLt")
    ap.add_argument("--optimizer", default="flat", choices=["flat", "torch"], help="flat = ShardedFlatAdam (fused kernel)")
    a = ap.parse_args()
    torch.manual_seed(0)
    model = build(a.model, a.dropout).train()
    if a.optimizer == "flat":
        from adafortitran_amd.optim import ShardedFlatAdam
        opt = ShardedFlatAdam(model.parameters(), lr=1e-3)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    inp = synth.make_inputs(a.batch, seed=1)
    pil, tgt = torch.from_numpy(inp["pilots"]).cuda(), torch.from_numpy(inp["target"]).cuda()
    meta = synth.meta_tuple(inp) if a.model == "adafortitran" else None

    def step():
        opt.zero_grad()
        out = model(pil, meta) if meta is not None else model(pil)
        loss = torch.nn.functional.mse_loss(torch.view_as_real(out), torch.view_as_real(tgt))
        loss.backward()
        opt.step()

    res = {}
    for mode in ("hip", "torch"):
        if a.only and a.only != mode:
            continue
        model.transformer_encoder.hip_training = mode == "hip"
        model.initial_enhancer.hip_training = model.final_refiner.hip_training = mode == "hip"
        training.HipLinear.default_hip_training = mode == "hip" and a.dense == "hip"
        if hasattr(model, "channel_adapter"):
            model.channel_adapter.hip_training = mode == "hip"
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / a.steps * 1e3
    out = {"metric": "training step (fwd+bwd+Adam) ms", "batch": a.batch, "model": a.model, "dropout": a.dropout,
           "ms_per_step": {k: round(v, 3) for k, v in res.items()},
           "frames_per_s": {k: round(a.batch / v * 1e3, 1) for k, v in res.items()}}
    if len(res) == 2:
        out["speedup_vs_pytorch_rocm"] = round(res["torch"] / res["hip"], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
