"""bench.py's launcher contract on a box without GPUs: `--gpus N` must never silently run one rank
(VERDICT r1 item 9).  The GPU legs are exercised by the driver's own bench run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, env=e, timeout=300)


def _visible_gpus():
    import torch
    return torch.cuda.device_count()


def test_gpus_n_without_enough_devices_exits_nonzero_and_prints_no_line():
    n = _visible_gpus() + 2
    res = _run(["--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0"])
    assert res.returncode != 0
    assert '"metric"' not in res.stdout
    assert "device(s) visible" in res.stderr


def test_world_size_mismatch_exits_nonzero():
    res = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and '"metric"' not in res.stdout
    assert "WORLD_SIZE=2" in res.stderr


def test_flop_accounting_matches_survey_8d():
    """SURVEY.md 8(d) per-frame figures: C3 1,429,387,932 FLOP/frame (adapter per plane) -- the kernel classes of
    bench.py cover everything but the adapter MLPs (73,290 MACs x 2 per plane = the difference)."""
    sys.path.insert(0, ROOT)
    import bench
    fl = bench.algorithmic_flops(bench.C3, 128)
    per_frame = fl["forward_total"] / 128
    adapter = 2 * 2 * (7 + 7 * 42 + 42 * 560)        # 2 planes x MAC=2 x (1*7 + 7*42 + 42*560) weights
    assert abs(per_frame + adapter - 1_429_387_932) / 1_429_387_932 < 2e-3
    assert fl["encoder_total"] / 128 == 1_362_493_440   # K6a-d dense-GEMM FLOPs per frame
    fl5 = bench.algorithmic_flops(bench.C5, 64)
    assert fl5["encoder_total"] / 64 == 59_013_857_280


@pytest.mark.gpu
def test_bench_headline_line_on_gpu():
    res = _run(["--steps", "5", "--warmup", "2", "--headline-only"])
    assert res.returncode == 0, res.stderr
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["unit"] == "frames/s" and line["value"] > 1000
    assert 0 < line["roofline"]["frac"] < 1
