"""Build the gfx950 shared library in-tree (adafortitran_amd/csrc/libaft_hip.so).

    python -m adafortitran_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the built .so is git-ignored but
travels with the tree to the GPU box.  gfx950 only: no other --offload-arch, no
compatibility layers.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ("aft_api.hip", "aft_train.hip", "k_chain.hip", "k_attn.hip", "k_encoder.hip", "k_conv.hip", "k_conv_stream.hip", "k_conv_rows.hip", "k_misc.hip", "k_gemm.hip",
           "k_attn_train.hip", "k_train.hip", "k_conv_train.hip", "k_chain_bwd.hip", "k_ends_train.hip")
LIB = os.path.join(CSRC, "libaft_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _deps(src: str):
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]
    return [os.path.join(CSRC, src), *headers, os.path.join(CSRC, "..", "..", "include", "adafortitran_amd.h")]


DIAG = False  # --diag: -DAFT_DIAG_STAMPS (in-kernel phase stamps; never for the product build)


def _compile(src: str, force: bool, objdir: str, extra) -> str:
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in _deps(src)):
        return obj
    flags = FLAGS + (["-DAFT_DIAG_STAMPS"] if DIAG else []) + list(extra)
    subprocess.run([HIPCC, *flags, "-c", os.path.join(CSRC, src), "-o", obj], check=True)
    return obj


CHECK_FLAGS = ("-DAFT_CHECKED=1",)   # the checked build (aft_internal.h: asserts, fenced hand-overs, bounded polls); never the product


def build_checked(force: bool = False, verbose: bool = False) -> str:
    """`libaft_hip_check.so`: the same sources with -DAFT_CHECKED=1, run against the product build by tests/test_checked_build.py."""
    return build(force=force, verbose=verbose, variant="check", extra_flags=CHECK_FLAGS)


def build(force: bool = False, verbose: bool = False, variant: str = "", extra_flags=()) -> str:
    """The product library (variant ""), or a variant `libaft_hip_<variant>.so` compiled with extra -D flags into its own object
    directory (tools/ab_kernels.py loads several of them in one process; the checked build is one).  A variant rebuilds when a source
    is newer than its objects or when its flags changed."""
    objdir, lib = CSRC, LIB
    if variant:
        objdir = os.path.join(CSRC, "build_" + variant)
        os.makedirs(objdir, exist_ok=True)
        lib = os.path.join(CSRC, f"libaft_hip_{variant}.so")
        stamp = os.path.join(objdir, "flags.txt")
        flags_now = " ".join(list(extra_flags) + (["-DAFT_DIAG_STAMPS"] if DIAG else []))
        if not os.path.exists(stamp) or open(stamp).read() != flags_now:
            force = True
            with open(stamp, "w") as fh:
                fh.write(flags_now)
    with ThreadPoolExecutor(max_workers=4) as pool:
        objs = list(pool.map(lambda s: _compile(s, force, objdir, extra_flags), SOURCES))
    if force or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs,
                        "-Wl,-rpath,/opt/rocm/lib"], check=True)
        if verbose:
            print("built", lib)
    return lib


if __name__ == "__main__":
    # python -m adafortitran_amd.build [--force] [--diag] [--variant NAME -DX=1 -DY=2 ...]      (--variant check -DAFT_CHECKED=1 = build_checked)
    DIAG = "--diag" in sys.argv
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    build(force="--force" in sys.argv or DIAG, verbose=True, variant=var, extra_flags=[a for a in sys.argv[1:] if a.startswith("-D")])
