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


def _dp_worker(rank, world, port, out):
    """Data-parallel training of the whole AdaFortiTran on the HIP training kernels, two ranks SHARING device 0 with the
    collectives on gloo (RCCL refuses two ranks on one device): each rank trains on its half of the frames."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out[rank] = _dp_train(rank, world)
    dist.barrier()
    dist.destroy_process_group()


def _dp_model_and_data(frames):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import train_bench
    from adafortitran_amd import synth
    torch.manual_seed(0)
    model = train_bench.build("adafortitran", 0.0).train()      # dropout 0: the ranks' masks would differ from one process's
    inp = synth.make_inputs(frames, seed=21)
    pil = torch.from_numpy(inp["pilots"]).cuda()
    meta = tuple(m.cuda() if hasattr(m, "cuda") else m for m in synth.meta_tuple(inp))
    tgt = torch.from_numpy(inp["target"]).cuda()
    return model, pil, meta, tgt


def _dp_train(rank, world, steps=3):
    from adafortitran_amd.optim import ShardedFlatAdam
    frames = 8
    steps = int(os.environ.get("AFT_DP_STEPS", steps))          # (tools/debug/dp_two_rank_diff.py looks at one and two steps)
    model, pil, meta, tgt = _dp_model_and_data(frames)
    lo, hi = rank * frames // world, (rank + 1) * frames // world
    opt = ShardedFlatAdam(model.parameters(), lr=1e-3) if world > 1 else torch.optim.Adam(model.parameters(), lr=1e-3)
    first_grad = None
    for it in range(steps):
        opt.zero_grad()
        est = model(pil[lo:hi], tuple(m[lo:hi] for m in meta))
        torch.nn.functional.mse_loss(torch.view_as_real(est), torch.view_as_real(tgt[lo:hi])).backward()   # equal shards: mean of means
        if it == 0:
            if world > 1:
                opt.reduce_gradients()        # the averaged gradient in every .grad, as DDP leaves it; step() then skips its own reduction
            first_grad = torch.cat([p.grad.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
        opt.step()
    torch.cuda.synchronize()
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy(), first_grad


def _two_ranks_and_one_process(steps):
    os.environ["AFT_DP_STEPS"] = str(steps)
    try:
        port = _free_port()
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
            got = dict(ret)
        want, want_grad = _dp_train(0, 1)
    finally:
        os.environ.pop("AFT_DP_STEPS", None)
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])      # the ranks hold the same bits
    return got[0][0], got[0][1], want, want_grad


def test_two_ranks_sharing_the_gpu_train_like_one_process():
    """SURVEY 8e/8f-1 on hardware at world size 2, as far as a 1-GPU box allows: frames sharded over the ranks, every rank's
    forward + backward on the HIP training kernels, ShardedFlatAdam's reduce-scatter -> fused Adam on the shard -> all-gather
    on device buffers.  What must hold: both ranks keep bit-identical parameters; the ranks' averaged first gradient equals ONE
    process's gradient on all the frames to 1e-5 of its maximum (fp32 sums in another order; measured 2e-8); after ONE step the
    parameters agree to 1 % of lr (Adam's g / (sqrt(v) + eps) at |g| ~ eps = 1e-8 turns a relative 1e-4 of rounding noise into a
    fraction of a percent; measured 0.3 %, all but one in ten thousand within 1e-4 of lr).  After three steps the two runs are two
    trajectories of a chaotic map started 3e-6 apart -- the distance grows ~10 x per step whatever the kernels (measured maxima after
    1 / 2 / 3 steps: 2.9e-6 / 9.2e-5 / 2.7e-4; a forward whose roundings differ in the last bit starts the same map somewhere else:
    9.7e-6 / 1.3e-4 / 8.1e-4 was measured with another bias-add order in the embedding; `tools/debug/dp_two_rank_diff.py` lists
    the tensors) -- so that comparison is loose by construction: all but one parameter in ten thousand within 5 % of the distance
    travelled (3 steps x lr; measured 3.2 %), none beyond 20 % (measured 9 %)."""
    lr = 1e-3
    par, grad, want, want_grad = _two_ranks_and_one_process(1)
    gerr = np.abs(grad - want_grad).max()
    assert gerr <= 1e-5 * np.abs(want_grad).max(), (gerr, np.abs(want_grad).max())
    d1 = np.abs(par - want)
    assert np.quantile(d1, 0.9999) <= 1e-3 * lr and d1.max() <= 0.01 * lr, (np.quantile(d1, 0.9999), d1.max())
    par3, _, want3, _ = _two_ranks_and_one_process(3)
    d3 = np.abs(par3 - want3)
    assert np.quantile(d3, 0.9999) <= 0.05 * 3 * lr and d3.max() <= 0.2 * 3 * lr, (np.quantile(d3, 0.9999), d3.max())
