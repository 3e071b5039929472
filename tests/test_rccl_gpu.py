"""RCCL gets a real execution: MseAccumulator.result() and one ShardedFlatAdam.step()
under backend "nccl" (= RCCL on ROCm) on device buffers -- world_size 2 when two devices are visible, else
world_size 1 on cuda:0 (still loads RCCL and runs all_gather / reduce_scatter_tensor /
all_gather_into_tensor through it).  Each rank runs in its own process, as the product does."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from adafortitran_amd import synth
    from adafortitran_amd.metrics import MseAccumulator
    from adafortitran_amd.optim import ShardedFlatAdam
    # ---- metric closure: every rank accumulates its own shard, one all-gather of the 16-byte pairs ----
    inp = synth.make_inputs(6, seed=11)
    est = torch.from_numpy(inp["target"]).to(dev)
    ref = torch.from_numpy(inp["target"] * np.complex64(0.5)).to(dev)
    lo, hi = (6 * rank) // world, (6 * (rank + 1)) // world
    acc = MseAccumulator(dev)
    acc.update(est[lo:hi], ref[lo:hi])
    mse = acc.result()
    want = float(np.mean(np.abs(inp["target"] * 0.5) ** 2, dtype=np.float64))
    # ---- one optimizer step: reduce_scatter(grads) -> fused Adam on the shard -> all_gather(params) ----
    torch.manual_seed(0)
    lin = torch.nn.Linear(40, 24).to(dev)
    twin = torch.nn.Linear(40, 24).to(dev)
    twin.load_state_dict(lin.state_dict())
    opt = ShardedFlatAdam(lin.parameters(), lr=1e-2)
    ref_opt = torch.optim.Adam(twin.parameters(), lr=1e-2)
    x = torch.randn(16, 40, device=dev)                # same seed on every rank -> identical local gradients,
    for m, o in ((lin, opt), (twin, ref_opt)):         # so the mean over ranks equals the single-rank gradient
        o.zero_grad()
        m(x).square().mean().backward()
        o.step()
    torch.cuda.synchronize()
    err = max(float((a - b).abs().max()) for a, b in zip(lin.parameters(), twin.parameters()))
    ret[rank] = (mse, want, err, opt.world)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_allgather_metric_and_sharded_adam_step():
    world = 2 if torch.cuda.device_count() >= 2 else 1
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        results = dict(ret)
    assert len(results) == world
    for rank in range(world):
        mse, want, err, w = results[rank]
        assert w == world
        assert abs(mse - want) <= 1e-6 * want          # global value on every rank
        assert err <= 1e-6                             # sharded step == torch.optim.Adam step
