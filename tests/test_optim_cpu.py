"""Flat-buffer Adam and the sharded data-parallel step on CPU: single process against
torch.optim.Adam, and world_size 2 over gloo against the single-process result on the full batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from adafortitran_amd.optim import ShardedFlatAdam


def _net(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 17), torch.nn.GELU(), torch.nn.Linear(17, 3))


def _data(n=8, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 6, generator=g), torch.randn(n, 3, generator=g)


def test_flat_adam_matches_torch_adam():
    a, b = _net(), _net()
    x, y = _data()
    ref = torch.optim.Adam(a.parameters(), lr=1e-2, weight_decay=1e-3)
    opt = ShardedFlatAdam(b.parameters(), lr=1e-2, weight_decay=1e-3)
    for _ in range(5):
        for net, o in ((a, ref), (b, opt)):
            o.zero_grad()
            torch.nn.functional.mse_loss(net(x), y).backward()
            o.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7)


def test_scheduler_clipping_and_checkpoint():
    """ExponentialLR drives the lr (reference trainer.py:414); max_grad_norm reproduces clip_grad_norm_ +
    Adam; state_dict round-trips the moments."""
    a, b = _net(), _net()
    x, y = _data()
    ref = torch.optim.Adam(a.parameters(), lr=1e-2)
    opt = ShardedFlatAdam(b.parameters(), lr=1e-2, max_grad_norm=0.05)
    sa = torch.optim.lr_scheduler.ExponentialLR(ref, gamma=0.9)
    sb = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.9)
    for _ in range(4):
        ref.zero_grad(); opt.zero_grad()
        torch.nn.functional.mse_loss(a(x), y).backward()
        torch.nn.utils.clip_grad_norm_(a.parameters(), 0.05)
        ref.step(); sa.step()
        torch.nn.functional.mse_loss(b(x), y).backward()
        opt.step(); sb.step()
    assert abs(opt.param_groups[0]["lr"] - ref.param_groups[0]["lr"]) < 1e-12
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7)
    sd = opt.state_dict()
    c = _net()
    opt2 = ShardedFlatAdam(c.parameters(), lr=1.0)
    opt2.load_state_dict(sd)
    assert opt2.steps == 4 and torch.equal(opt2.exp_avg, opt.exp_avg) and opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]


def test_state_dict_is_exchangeable_with_torch_adam():
    """The checkpoint is torch.optim.Adam's format: a ShardedFlatAdam state loads into torch.optim.Adam over
    the same parameters and the two continue identically; and the other way round."""
    a, b = _net(), _net()
    x, y = _data()
    opt = ShardedFlatAdam(a.parameters(), lr=1e-2)
    ref = torch.optim.Adam(b.parameters(), lr=1e-2)
    for _ in range(3):
        for net, o in ((a, opt), (b, ref)):
            o.zero_grad()
            torch.nn.functional.mse_loss(net(x), y).backward()
            o.step()
    c, d = _net(), _net()
    c.load_state_dict(a.state_dict()); d.load_state_dict(b.state_dict())
    into_torch = torch.optim.Adam(c.parameters(), lr=1.0)
    into_torch.load_state_dict(opt.state_dict())               # ours -> torch
    into_flat = ShardedFlatAdam(d.parameters(), lr=1.0)
    into_flat.load_state_dict(ref.state_dict())                # torch -> ours
    assert into_flat.steps == 3 and into_torch.param_groups[0]["lr"] == 1e-2
    for net, o in ((c, into_torch), (d, into_flat), (a, opt)):
        o.zero_grad()
        torch.nn.functional.mse_loss(net(x), y).backward()
        o.step()
    for p, q, r in zip(a.parameters(), c.parameters(), d.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7) and torch.allclose(p, r, rtol=1e-5, atol=1e-7)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _net()
    x, y = _data()
    lo, hi = rank * len(x) // world, (rank + 1) * len(x) // world
    opt = ShardedFlatAdam(net.parameters(), lr=1e-2)
    for _ in range(4):
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]).backward()   # equal shards: mean of means = global mean
        opt.step()
    out[rank] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    dist.destroy_process_group()


def _scaler_worker(rank, world, port, out):
    """The reference's GradScaler branch on two ranks with an inf injected into ONE rank's gradients: with
    reduce_gradients() both ranks see the inf, both skip, nobody hangs, parameters stay identical; without it the
    optimizer refuses instead of desynchronising the collectives."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _net()
    x, y = _data()
    lo, hi = rank * len(x) // world, (rank + 1) * len(x) // world
    opt = ShardedFlatAdam(net.parameters(), lr=1e-2)
    scaler = torch.amp.GradScaler("cpu", init_scale=4.0, growth_interval=1000)
    before = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    log = []
    for it in range(3):
        opt.zero_grad()
        scaler.scale(torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi])).backward()
        if it == 1 and rank == 1:
            next(net.parameters()).grad.view(-1)[0] = float("inf")      # only rank 1 overflows
        if it == 0:
            try:
                scaler.step(opt)                                         # unreduced grads on 2 ranks: refused
                log.append("stepped")
            except RuntimeError as exc:
                log.append("refused" if "reduce_gradients" in str(exc) else "other")
            scaler = torch.amp.GradScaler("cpu", init_scale=4.0, growth_interval=1000)
            continue
        opt.reduce_gradients()
        scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
        scaler.step(opt)
        scaler.update()
        log.append((opt.steps, float(scaler.get_scale())))
    after = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    out[rank] = (log, after.numpy(), bool(torch.equal(before, after)))
    dist.destroy_process_group()


def test_grad_scaler_with_injected_inf_on_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_scaler_worker, args=(2, port, out), nprocs=2, join=True)
    (log0, p0, same0), (log1, p1, same1) = out[0], out[1]
    assert log0[0] == "refused" and log1[0] == "refused"
    # iteration 1: inf on rank 1 only -> BOTH ranks skip (steps stays 0) and both back the scale off 4 -> 2
    assert log0[1] == (0, 2.0) and log1[1] == (0, 2.0)
    # iteration 2: clean -> both step once
    assert log0[2][0] == 1 and log1[2][0] == 1
    assert np.array_equal(p0, p1) and not same0 and np.isfinite(p0).all()


def test_sharded_step_world_size_2_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    net = _net()
    x, y = _data()
    ref = torch.optim.Adam(net.parameters(), lr=1e-2)
    for _ in range(4):
        ref.zero_grad()
        torch.nn.functional.mse_loss(net(x), y).backward()
        ref.step()
    want = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    assert np.allclose(out[0], out[1], rtol=0, atol=0)          # ranks hold identical parameters
    assert np.allclose(out[0], want, rtol=1e-5, atol=1e-7)


def test_sharded_step_world_size_8_matches_single_process():
    """The node size of BASELINE configs 4 / 5: eight ranks, one sample each; the flat buffers are padded to 8 x 64 elements so that
    every rank owns an equal shard (173 parameters -> 512 padded, 64 per rank: ranks 3..7 own padding only and must still take part
    in both collectives)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(8, port, out), nprocs=8, join=True)
    net = _net()
    x, y = _data()
    ref = torch.optim.Adam(net.parameters(), lr=1e-2)
    for _ in range(4):
        ref.zero_grad()
        torch.nn.functional.mse_loss(net(x), y).backward()
        ref.step()
    want = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    assert all(np.array_equal(out[0], out[r]) for r in range(1, 8))      # ranks hold identical parameters
    assert np.allclose(out[0], want, rtol=1e-5, atol=1e-7)


def test_shard_layout_at_world_size_8():
    """Shard arithmetic without a process group: padded length divisible by 8 x 64, shards tile it exactly."""
    from adafortitran_amd.optim import FlatParameters
    net = _net()
    flat = FlatParameters(net.parameters(), pad_to=8 * 64)
    n = sum(p.numel() for p in net.parameters())
    assert flat.padded % (8 * 64) == 0 and flat.padded >= n
    shard = flat.padded // 8
    assert [(r * shard, (r + 1) * shard) for r in range(8)][-1][1] == flat.padded
    flat.release()


def test_flat_parameters_do_not_break_pickling_and_double_reduce_raises():
    """the ownership tag on a Parameter is a picklable token (torch.save(model) / mp.spawn used to fail on a
    weakref while an optimizer was alive); reduce_gradients() twice in one step raises instead of mixing averaged and
    rank-local gradients."""
    import io
    import pickle

    import pytest

    from adafortitran_amd.optim import flat_owner
    net = _net()
    opt = ShardedFlatAdam(net.parameters(), lr=1e-2)
    buf = io.BytesIO()
    torch.save(net, buf)                                   # pickles Parameter.__dict__
    pickle.dumps(list(net.parameters()))
    assert all(flat_owner(p) is opt.flat for p in net.parameters())
    buf.seek(0)
    clone = torch.load(buf, weights_only=False)
    assert all(flat_owner(p) is opt.flat or flat_owner(p) is None for p in clone.parameters())
    x, y = _data()
    torch.nn.functional.mse_loss(net(x), y).backward()
    opt.reduce_gradients()
    with pytest.raises(RuntimeError, match="already called"):
        opt.reduce_gradients()
    v0 = [p._version for p in net.parameters()]
    opt.step()
    assert all(p._version > v for p, v in zip(net.parameters(), v0))   # raw updates move the version counters
    opt.zero_grad()
    torch.nn.functional.mse_loss(net(x), y).backward()
    opt.reduce_gradients()                                  # a new step may reduce again
    opt.flat.release()
    assert all(flat_owner(p) is None for p in net.parameters())
