"""Drop-in check against the reference's OWN training scaffolding (INTEGRATION.md level 1), on CPU:
the reference's ``TrainingLoop`` / ``ModelEvaluator`` / ``MatDataset`` (imported from /root/reference,
untouched) drive THIS package's ``src.models`` / ``src.config`` shims on a small synthetic dataset.

Runs only where the reference tree is mounted (never on the GPU box).  The overlay ``src`` package is
built from symlinks in a temp dir; ``tensorboard`` / ``prettytable`` (absent from this image, used by the
reference only for logging) are replaced by no-op modules inside the child process."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import scipy.io as sio

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "main")), reason="reference tree not mounted")


def _write_dataset(root, n_train=12, n_val=6):
    rng = np.random.default_rng(0)
    S, T, ps, pt = 120, 14, 12, 2
    rows, cols = np.arange(0, S, S // ps)[:ps], np.array([3, 10])

    def one(path, idx, snr):
        h = (rng.standard_normal((S, T)) + 1j * rng.standard_normal((S, T))).astype(np.complex64)
        sparse = np.zeros((S, T), np.complex64)
        sparse[np.ix_(rows, cols)] = h[np.ix_(rows, cols)] + 0.05 * (rng.standard_normal((ps, pt)) + 1j * rng.standard_normal((ps, pt)))
        H = np.stack([h, sparse, h], axis=2).astype(np.complex128)
        sio.savemat(os.path.join(path, f"{idx}_SNR-{snr}_DS-100_DOP-400_N-3_TDL-A.mat"), {"H": H})

    for split, n in (("train", n_train), ("val", n_val)):
        os.makedirs(os.path.join(root, split))
        for i in range(n):
            one(os.path.join(root, split), i, 10)
    for snr in (0, 20):
        d = os.path.join(root, "test", "SNR_test_set", f"SNR_{snr}")
        os.makedirs(d)
        for i in range(3):
            one(d, i, snr)


CHILD = textwrap.dedent("""
    import sys, types, os, json
    sys.dont_write_bytecode = True
    ovl, repo, data = sys.argv[1], sys.argv[2], sys.argv[3]
    sys.path[:0] = [ovl, repo]
    import typing
    if not hasattr(typing, "Self"):
        import typing_extensions
        typing.Self = typing_extensions.Self
    for name in ("torch.utils.tensorboard", "torch.utils.tensorboard.writer"):
        m = types.ModuleType(name)
        m.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda s, *a, **k: None, "add_scalar": lambda s, *a, **k: None,
                                                      "close": lambda s: None})
        sys.modules[name] = m
    pt = types.ModuleType("prettytable")
    pt.PrettyTable = type("PrettyTable", (), {"__init__": lambda s, *a, **k: None, "add_row": lambda s, *a, **k: None})
    sys.modules["prettytable"] = pt
    import logging, torch
    from torch import nn, optim
    from torch.utils.data import DataLoader
    import src.main.trainer as T                      # the reference's trainer module, as shipped
    from src.data import MatDataset, get_test_dataloaders
    from src.models import AdaFortiTranEstimator      # resolves to this package through the shim
    from src.config.schemas import SystemConfig, ModelConfig
    import adafortitran_amd
    assert AdaFortiTranEstimator is adafortitran_amd.AdaFortiTranEstimator
    assert os.path.realpath(T.__file__).startswith("/root/reference/")
    torch.manual_seed(0)
    sc = SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    mc = ModelConfig(model_type="adafortitran", patch_size=(3, 2), num_layers=1, model_dim=32, num_head=2, activation="gelu",
                     dropout=0.1, max_seq_len=512, pos_encoding_type="learnable", device="cpu",
                     channel_adaptivity_hidden_sizes=[7, 42, 560], adaptive_token_length=6)
    model = AdaFortiTranEstimator(sc, mc)
    opt = optim.Adam(model.parameters(), lr=2e-3)
    sched = optim.lr_scheduler.ExponentialLR(opt, gamma=0.995)
    loop = T.TrainingLoop(model, opt, sched, nn.MSELoss(), torch.device("cpu"), gradient_clip_val=1.0)
    train = DataLoader(MatDataset(os.path.join(data, "train"), sc.pilot), batch_size=4, shuffle=True)
    val = DataLoader(MatDataset(os.path.join(data, "val"), sc.pilot), batch_size=4)
    losses = [loop.train_epoch(train) for _ in range(4)]
    vloss = loop.evaluate(val)
    ev = T.ModelEvaluator(model, torch.device("cpu"), logging.getLogger("t"))
    tests = get_test_dataloaders(os.path.join(data, "test", "SNR_test_set"), sc.pilot, 2)
    stats = ev.get_test_stats(tests, nn.MSELoss())
    print(json.dumps({"train": losses, "val": vloss, "test": {str(k): v for k, v in stats.items()}}))
""")


def test_reference_training_loop_drives_this_package(tmp_path):
    ovl = tmp_path / "ovl" / "src"
    ovl.mkdir(parents=True)
    (ovl / "__init__.py").write_text("")
    for name, src in (("models", os.path.join(REPO, "src", "models")), ("config", os.path.join(REPO, "src", "config")),
                      ("main", os.path.join(REF, "src", "main")), ("data", os.path.join(REF, "src", "data")),
                      ("utils.py", os.path.join(REF, "src", "utils.py"))):
        os.symlink(src, ovl / name)
    data = tmp_path / "data"
    _write_dataset(str(data))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, "-c", CHILD, str(tmp_path / "ovl"), REPO, str(data)], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert all(np.isfinite(out["train"])) and out["train"][-1] < out["train"][0]
    assert np.isfinite(out["val"]) and set(out["test"]) == {"0", "20"}
