"""Ingest (f2), fused evaluation sweep (f3) and LS baseline (f4): host logic + numpy oracle against
golden values produced by the reference's MatDataset / extract_values / get_ls_mse_per_folder
(tests/golden/I_ingest.npz), and the GPU kernels against the same goldens."""
import os

import numpy as np
import pytest
import scipy.io as sio
import torch

from adafortitran_amd import ingest
from adafortitran_amd.evaluation import evaluate_dataloader, get_test_stats
from helpers import GOLDEN, Golden

I = np.load(os.path.join(GOLDEN, "I_ingest.npz"))


def _write_tree(root):
    """Re-create the synthetic .mat tree the golden file was generated from."""
    for key in I.files:
        if key.startswith("H__"):
            _, folder, fname = key.split("__")
            os.makedirs(os.path.join(root, folder), exist_ok=True)
            sio.savemat(os.path.join(root, folder, fname), {"H": I[key].astype(np.complex128)})


def test_oracle_restatements_match_reference():
    from oracle import ingest_oracle as io
    for i, name in enumerate(I["names"]):
        folder, fname = str(name).split("/")
        H = I[f"H__{folder}__{fname}"]
        hp, hi = io.process_channel_data(H, (12, 2))
        assert np.array_equal(hp, I["pilots"][i]) and np.array_equal(hi, I["ideal"][i])
        assert list(io.extract_values(fname)[:5]) == list(I["meta"][i])
    for key, val in zip(I["ls_keys"], I["ls_vals"]):
        per_file = [io.mse_db(I[k][:, :, 2].astype(np.complex128), I[k][:, :, 0].astype(np.complex128))
                    for k in I.files if k.startswith(f"H__SNR_{key}__")]
        assert abs(np.mean(per_file) - val) <= 1e-5


def test_parse_filename_and_errors():
    assert ingest.parse_filename("1_SNR-20_DS-50_DOP-500_N-3_TDL-A.mat") == (1.0, 20.0, 50.0, 500.0, 3.0, "TDL-A")
    with pytest.raises(ValueError, match="Cannot extract file information"):
        ingest.parse_filename("sample.mat")


def test_packed_loader_matches_reference_dataset(tmp_path):
    _write_tree(str(tmp_path))
    packed = ingest.pack_mat_folder(tmp_path / "SNR_10", tmp_path / "snr10.npz")
    loader = ingest.PackedLoader(str(tmp_path / "snr10.npz"), (12, 2), batch_size=2)
    got_p, got_h, got_m = [], [], []
    for pilots, ideal, meta in loader:
        assert pilots.dtype == torch.complex64 and ideal.dtype == torch.complex64 and len(meta) == 6
        assert meta[1].shape == (pilots.shape[0], 1) and len(meta[5][0]) == pilots.shape[0]
        got_p.append(pilots.numpy()); got_h.append(ideal.numpy()); got_m.append(torch.cat(meta[:5], dim=1).numpy())
    sel = [i for i, n in enumerate(I["names"]) if str(n).startswith("SNR_10/")]
    assert np.array_equal(np.concatenate(got_p), I["pilots"][sel])
    assert np.array_equal(np.concatenate(got_h), I["ideal"][sel])
    assert np.array_equal(np.concatenate(got_m), I["meta"][sel])
    assert packed["h_ideal"].shape == (3, 120, 14)
    bad = dict(packed)
    bad["h_ls_sparse"] = packed["h_ls_sparse"].copy()
    bad["h_ls_sparse"][1, 0, 0] = 1.0                         # a 25th non-zero entry
    with pytest.raises(ValueError, match="Expected 24 pilot values, got 25"):
        list(ingest.PackedLoader(bad, (12, 2), 4))


def test_ls_baseline_cpu(tmp_path):
    _write_tree(str(tmp_path))
    got = ingest.get_ls_mse_per_folder(str(tmp_path))
    assert list(got) == sorted(int(k) for k in I["ls_keys"])
    for key, val in zip(I["ls_keys"], I["ls_vals"]):
        assert abs(got[int(key)] - val) <= 1e-4


def test_eval_sweep_matches_reference_formula_cpu(tmp_path):
    from test_estimators_cpu import build_model
    g = Golden("D_forti")
    model = build_model(g)
    batches = [(torch.from_numpy(g["pilots"][i:i + 3]), torch.from_numpy(g["target"][i:i + 3]), None)
               for i in range(0, 8, 3)]
    mse = evaluate_dataloader(model, batches)
    assert abs(mse - g.meta["metric_2xmse"]) <= 1e-5 * g.meta["metric_2xmse"]
    stats = get_test_stats(model, [("SNR_20", batches), ("SNR_5", batches[:1])])
    assert list(stats) == [5, 20] and abs(stats[20] - g.meta["metric_db"]) <= 1e-4


@pytest.mark.gpu
def test_pilot_gather_and_ls_mse_kernels(tmp_path):
    from adafortitran_amd.hip_ops import ls_mse_db, pilot_gather
    keys = [f"H__{str(n).split('/')[0]}__{str(n).split('/')[1]}" for n in I["names"]]
    H = np.stack([I[k] for k in keys])
    sparse = torch.from_numpy(np.ascontiguousarray(H[:, :, :, 1])).cuda()
    pilots = pilot_gather(sparse, (12, 2)).cpu().numpy()
    assert np.array_equal(pilots, I["pilots"])                 # bit-exact, order preserved
    big = sparse.repeat(40, 1, 1)                              # 200 frames: many workgroups
    assert np.array_equal(pilot_gather(big, (12, 2)).cpu().numpy(), np.tile(I["pilots"], (40, 1, 1)))
    sparse[2, 5, 5] = 1.0
    with pytest.raises(ValueError, match=r"Expected 24 pilot values, got 25 \(frame 2\)"):
        pilot_gather(sparse, (12, 2))
    # a frame with FEWER non-zero entries than expected: the kernel itself zero-fills the rest of that frame's output (the output
    # tensor comes from torch.empty: poison the allocator first), and the count says 23
    from adafortitran_amd.hip_ops import pilot_gather as pg
    few = torch.from_numpy(np.ascontiguousarray(H[:, :, :, 1])).cuda()
    pos = (few[1] != 0).nonzero()[-1]
    few[1, pos[0], pos[1]] = 0
    junk = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 20, 1 << 16, 1 << 12, 1 << 10)]
    del junk
    got, counts = pg(few, (12, 2), return_counts=True)
    want = I["pilots"].copy()
    want[1].reshape(-1)[-1] = 0
    assert np.array_equal(got.cpu().numpy(), want) and counts.cpu().tolist()[1] == 23
    from oracle import ingest_oracle as io
    ls = torch.from_numpy(np.ascontiguousarray(H[:, :, :, 2])).cuda()
    ideal = torch.from_numpy(np.ascontiguousarray(H[:, :, :, 0])).cuda()
    db = ls_mse_db(ls, ideal).cpu().numpy()
    want = np.array([io.mse_db(H[i, :, :, 2].astype(np.complex128), H[i, :, :, 0].astype(np.complex128)) for i in range(len(H))])
    assert np.abs(db - want).max() <= 1e-4
    _write_tree(str(tmp_path))
    got = ingest.get_ls_mse_per_folder(str(tmp_path), device="cuda")
    for key, val in zip(I["ls_keys"], I["ls_vals"]):
        assert abs(got[int(key)] - val) <= 1e-4


@pytest.mark.gpu
def test_packed_loader_feeds_hip_model_end_to_end(tmp_path):
    """.mat tree -> packed file -> GPU pilot gather -> HIP forward -> device metric, against the
    CPU composite fed by the host path."""
    import adafortitran_amd as A
    from test_estimators_cpu import _configs
    g = Golden("A_ada")
    _write_tree(str(tmp_path))
    ingest.pack_mat_folder(tmp_path / "SNR_10", tmp_path / "p.npz")
    sd = {k: torch.from_numpy(v) for k, v in g.state_dict().items()}
    res = {}
    for dev in ("cpu", "cuda"):
        sc, mc = _configs(g.spec, device=dev)
        model = A.AdaFortiTranEstimator(sc, mc)
        model.load_state_dict(sd)
        res[dev] = evaluate_dataloader(model, ingest.PackedLoader(str(tmp_path / "p.npz"), (12, 2), 2, device=dev))
    assert abs(res["cuda"] - res["cpu"]) <= 1e-4 * res["cpu"]


@pytest.mark.gpu
def test_packed_loader_on_device_matches_host_and_raises_late():
    """PackedLoader on the HIP device (pinned grids, side-stream copies, GPU gather, counts checked one batch late):
    same batches as the host path bit for bit; a frame with 25 non-zero entries raises the reference's error with the
    frame's index in the folder -- while the NEXT batch is being prepared or at the end of the sweep, never silently."""
    rng = np.random.default_rng(5)
    N, S, T = 37, 120, 14
    ideal = (rng.standard_normal((N, S, T)) + 1j * rng.standard_normal((N, S, T))).astype(np.complex64)
    sparse = np.zeros((N, S, T), np.complex64)
    rows, cols = np.arange(0, S, 10), np.array([3, 10])
    sparse[:, rows[:, None], cols[None, :]] = ideal[:, rows[:, None], cols[None, :]]
    meta = rng.uniform(0, 30, (N, 5)).astype(np.float32)
    packed = {"h_ideal": ideal, "h_ls_sparse": sparse, "meta": meta, "channel_type": np.array(["TDL-A"] * N)}
    host = list(ingest.PackedLoader(packed, (12, 2), 8, device="cpu"))
    dev = list(ingest.PackedLoader(packed, (12, 2), 8, device="cuda"))
    assert len(host) == len(dev) == 5
    for (ph, ih, mh), (pd, idv, md) in zip(host, dev):
        assert torch.equal(pd.cpu(), ph) and torch.equal(idv.cpu(), ih)
        assert all(torch.equal(a, b) for a, b in zip(mh[:5], md[:5])) and mh[5] == md[5]
    for bad_frame in (3, 20, 36):          # first batch, a middle batch, the last (ragged) batch
        broken = dict(packed)
        broken["h_ls_sparse"] = sparse.copy()
        broken["h_ls_sparse"][bad_frame, 5, 5] = 1.0
        with pytest.raises(ValueError, match=rf"Expected 24 pilot values, got 25 \(frame {bad_frame}\)"):
            for _ in ingest.PackedLoader(broken, (12, 2), 8, device="cuda"):
                pass
    # a consumer that STOPS iterating right after the bad batch still gets the error (generator close)
    broken = dict(packed)
    broken["h_ls_sparse"] = sparse.copy()
    broken["h_ls_sparse"][2, 5, 5] = 1.0
    it = iter(ingest.PackedLoader(broken, (12, 2), 8, device="cuda"))
    next(it)                                               # the bad batch was yielded; its counts are still in flight
    with pytest.raises(ValueError, match=r"Expected 24 pilot values, got 25 \(frame 2\)"):
        it.close()
    # an exception from the consumer's own loop body propagates UNCHANGED (the pending count error does not replace it)
    class Boom(Exception):
        pass
    with pytest.raises(Boom):
        for _ in ingest.PackedLoader(broken, (12, 2), 8, device="cuda"):
            raise Boom()
    # that abandoned sweep's outstanding count error is not lost in an "Exception ignored in generator": it is
    # logged when the iterator is finalised and kept on the loader
    import gc
    loader = ingest.PackedLoader(broken, (12, 2), 8, device="cuda")
    with pytest.raises(ValueError, match=r"Expected 24 pilot values, got 25 \(frame 2\)"):
        with loader:
            for _ in loader:
                break
            gc.collect()
    loader.raise_pending()                                 # delivered once
    # packs above max_pinned_bytes go through the pinned ring instead of pinning the whole pack: same batches
    ring = list(ingest.PackedLoader(packed, (12, 2), 8, device="cuda", max_pinned_bytes=0))
    for (ph, ih, mh), (pd, idv, md) in zip(host, ring):
        assert torch.equal(pd.cpu(), ph) and torch.equal(idv.cpu(), ih)
