#!/usr/bin/env python3
"""Instruction mix of the basic blocks of one kernel in a hipcc --save-temps .s file: per block MFMA / vector / DPP / mov / LDS / waits.
usage: isa_loop_stats.py file.s mangled-name-substring [min_mfma]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(tuple(["E", ")"])) or (l.startswith("_Z") and key in l and ":" in l))
blocks, cur, name = [], collections.Counter(), "entry"
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end") or t.startswith("s_endpgm"):
        blocks.append((name, cur)); break
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append((name, cur)); cur, name = collections.Counter(), m.group(1); continue
    if not t or t.startswith((";", ".")): continue
    op = t.split()[0]
    cur["all"] += 1
    if op.startswith("v_mfma"): cur["mfma"] += 1
    elif op.startswith("v_"):
        cur["valu"] += 1
        if "dpp" in t or "row_sh" in t or "wave_sh" in t: cur["dpp"] += 1
        if op.startswith(("v_mov", "v_accvgpr")): cur["mov"] += 1
        if op.startswith("v_pk_"): cur["pk"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): cur["vmem"] += 1
    elif op.startswith("s_waitcnt"): cur["wait"] += 1
    elif op.startswith("s_nop"): cur["nop"] += 1
    elif op.startswith("s_"): cur["salu"] += 1
for name, c in blocks:
    if c["mfma"] >= min_mfma:
        print(name, dict(c))
