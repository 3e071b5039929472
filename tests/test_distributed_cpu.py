"""N>1 path on CPU: world_size-2 gloo processes shard the frames, run the estimator, and close
the sweep with the single all-gather of (sum|e|^2, n) -- must equal the 1-rank metric."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from adafortitran_amd import synth
from adafortitran_amd.metrics import MseAccumulator, shard_bounds, to_db
from helpers import Golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from test_estimators_cpu import build_model, golden_meta
    g = Golden(name)
    model = build_model(g)
    lo, hi = shard_bounds(g["pilots"].shape[0], world, rank)
    pil = torch.from_numpy(g["pilots"][lo:hi])
    tgt = torch.from_numpy(g["target"][lo:hi])
    meta = golden_meta(g)
    if meta is not None:
        meta = tuple(m[lo:hi] if torch.is_tensor(m) else m for m in meta)
    acc = MseAccumulator("cpu")
    with torch.no_grad():
        for i in range(0, hi - lo, 2):   # two frames per "batch": several updates per rank
            sl = slice(i, i + 2)
            m = tuple(t[sl] if torch.is_tensor(t) else t for t in meta) if meta is not None else None
            est = model(pil[sl], m) if m is not None else model(pil[sl])
            acc.update(est, tgt[sl])
    ret[rank] = (acc.result(), acc.n_elements)
    dist.destroy_process_group()


def test_two_rank_metric_equals_single_rank():
    name = "T_tiny_ada"
    g = Golden(name)
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, name, ret), nprocs=world, join=True)
        results = dict(ret)
    assert results[0][0] == results[1][0]                      # every rank holds the global value
    assert results[0][1] + results[1][1] == g["out"].size       # shards cover all frames exactly once
    assert abs(results[0][0] - g.meta["metric_2xmse"]) <= 1e-5 * g.meta["metric_2xmse"]
    assert abs(to_db(results[0][0]) - g.meta["metric_db"]) <= 1e-4


def test_eight_rank_metric_equals_single_rank():
    """World size 8 -- the node the scaling curve will be taken on: the fixture's 8 frames shard one per rank, the
    single all-gather carries eight (sum, n) pairs, every rank ends with the 1-rank metric."""
    name = "D_forti"
    g = Golden(name)
    world, port = 8, _free_port()
    assert g["pilots"].shape[0] >= world
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, name, ret), nprocs=world, join=True)
        results = dict(ret)
    assert len({results[r][0] for r in range(world)}) == 1     # every rank holds the global value, bit for bit
    assert sum(results[r][1] for r in range(world)) == g["out"].size
    assert all(results[r][1] > 0 for r in range(world))         # no rank idle at 8 frames over 8 ranks
    assert abs(results[0][0] - g.meta["metric_2xmse"]) <= 1e-5 * g.meta["metric_2xmse"]


def test_shard_bounds_cover_the_baseline_configs_at_world_size_8():
    """BASELINE configs 4 / 5: batch 1024 -> 128 per rank, batch 512 -> 64 per rank; ragged counts still partition exactly."""
    assert [shard_bounds(1024, 8, r) for r in range(8)] == [(128 * r, 128 * (r + 1)) for r in range(8)]
    assert [shard_bounds(512, 8, r) for r in range(8)] == [(64 * r, 64 * (r + 1)) for r in range(8)]
    for n in (1, 7, 9, 1023, 1025):
        cuts = [shard_bounds(n, 8, r) for r in range(8)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


def test_shard_bounds_partition():
    for n, w in ((1024, 8), (10, 3), (7, 8)):
        cuts = [shard_bounds(n, w, r) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))


def test_accumulator_matches_reference_metric_single_process():
    g = Golden("D_forti")
    acc = MseAccumulator("cpu")
    est, tgt = torch.from_numpy(g["out"]), torch.from_numpy(g["target"])
    acc.update(est[:3], tgt[:3])
    acc.update(est[3:], tgt[3:])
    assert abs(acc.result() - g.meta["metric_2xmse"]) <= 1e-6 * g.meta["metric_2xmse"]
    # the reference's own formula: 2 * MSELoss(cat(Re, Im; dim=1))  (utils.py:164-180)
    cat = lambda z: torch.cat((z.real, z.imag), dim=1)  # noqa: E731
    ref = 2.0 * torch.nn.MSELoss()(cat(est), cat(tgt)).item()
    assert abs(acc.result() - ref) <= 1e-6 * ref
