#!/usr/bin/env python3
"""Host time per enqueued step of the evaluation sweep's pieces (no device sync inside the timed loops): tells a
host-bound loop from a device-bound one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import adafortitran_amd as A
from adafortitran_amd import ingest, evaluation
from adafortitran_amd.metrics import MseAccumulator
N, B, S, T, PS, PT = 2048, 128, 120, 14, 12, 2
rng = np.random.default_rng(0)
ideal = (rng.standard_normal((N, S, T)) + 1j * rng.standard_normal((N, S, T))).astype(np.complex64)
sparse = np.zeros((N, S, T), np.complex64)
rows, cols = np.arange(0, S, S // PS)[:PS], np.array([3, 10])
sparse[:, rows[:, None], cols[None, :]] = ideal[:, rows[:, None], cols[None, :]]
meta = np.stack([rng.uniform(0, 30, N), rng.uniform(50, 350, N), rng.uniform(200, 1400, N), np.zeros(N), np.zeros(N)], 1).astype(np.float32)
packed = {"h_ideal": ideal, "h_ls_sparse": sparse, "meta": meta, "channel_type": np.array(["TDL-A"] * N)}
sc = A.SystemConfig(ofdm=dict(num_scs=S, num_symbols=T), pilot=dict(num_scs=PS, num_symbols=PT))
mc = A.ModelConfig(model_type="adafortitran", patch_size=(3, 2), num_layers=6, model_dim=128, num_head=4, activation="gelu",
                   max_seq_len=512, pos_encoding_type="learnable", device="cuda", dropout=0.1,
                   channel_adaptivity_hidden_sizes=[7, 42, 560], adaptive_token_length=6)
model = A.AdaFortiTranEstimator(sc, mc).eval()
loader = ingest.PackedLoader(packed, (PS, PT), B, device="cuda")
batches = [b for b in loader]
torch.cuda.synchronize()
acc = MseAccumulator(torch.device("cuda"))
with torch.no_grad():
    for name, fn in (("forward", lambda b: model(b[0], b[2])), ("forward+update", lambda b: acc.update(model(b[0], b[2]), b[1]))):
        for b in batches: fn(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches: fn(b)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{name}: host enqueue {t_host / len(batches) * 1e3:.3f} ms per batch, with device {t_all / len(batches) * 1e3:.3f} ms")
    for _ in range(2):
        t0 = time.perf_counter()
        n = 0
        for b in loader: n += 1
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
    print(f"loader: host {t_host / n * 1e3:.3f} ms per batch")
    t0 = time.perf_counter()
    for b in loader: acc.update(model(b[0], b[2]), b[1])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"sweep: host enqueue {t_host / n * 1e3:.3f} ms per batch, with device {t_all / n * 1e3:.3f} ms")
import cProfile, pstats
with torch.no_grad():
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        for b in batches: model(b[0], b[2])
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
