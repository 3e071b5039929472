"""CPU sanitizer run: the C restatement is rebuilt with
-fsanitize=address,undefined (oracle/Makefile `asan_driver`) and the tiny golden sets go through it in a child
process -- no sanitizer report, and the output still matches the reference-generated fixture."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from adafortitran_amd import _abi
from helpers import Golden, TOL_ORACLE_OUT

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(HERE), "oracle")


@pytest.fixture(scope="module")
def driver():
    res = subprocess.run(["make", "-C", ORACLE_DIR, "-s", "asan_driver"], capture_output=True, text=True)
    if res.returncode != 0:
        pytest.skip("no sanitizer-capable C toolchain here: " + res.stderr[-200:])
    return os.path.join(ORACLE_DIR, "asan_driver")


def _serialize(g: Golden, path: str) -> None:
    cfg, sd = g.abi_config(), g.state_dict()
    chunks, offsets, n = [], {}, 0

    def put(key, arr):
        nonlocal n
        a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        offsets[key] = n + 1                      # +1: 0 means NULL
        chunks.append(a)
        n += a.size
        return offsets[key]

    weights = _abi.make_weights(cfg, lambda k: put(k, sd[k]), pos_key=_abi.pos_key_of(sd))
    io = [put("pilots", g["pilots"].view(np.float32))]
    snr, ds, dop = g.meta_arrays()
    io += [put(k, v) if v is not None else 0 for k, v in (("snr", snr), ("ds", ds), ("dop", dop))]
    with open(path, "wb") as f:
        f.write(bytes(cfg))
        f.write(np.array([g["pilots"].shape[0], int(g.adaptive)], np.int32).tobytes())
        f.write(np.array([n], np.uint64).tobytes())
        f.write(np.concatenate(chunks).tobytes())
        table = bytes(weights._layer_table)               # the layer table travels behind the struct; its pointer slot is rebuilt by the driver
        weights.layers = None
        f.write(bytes(weights))
        f.write(table)
        f.write(np.array(io, np.uint64).tobytes())
    assert C.sizeof(weights) % 8 == 0


@pytest.mark.parametrize("name", ["T_tiny_ada", "T_tiny_forti"])
def test_oracle_under_asan_ubsan(driver, tmp_path, name):
    g = Golden(name)
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    _serialize(g, case)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="2")
    res = subprocess.run([driver, case, out], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-2000:]
    got = np.fromfile(out, dtype=np.float32).view(np.complex64).reshape(g["out"].shape)
    assert np.abs(got - g["out"]).max() <= TOL_ORACLE_OUT * max(1.0, np.abs(g["out"]).max())
