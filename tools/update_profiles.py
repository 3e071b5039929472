#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/ and derive
profiles/pmc_traffic.json (HBM bytes per launch of the dominant kernel, gfx950 corrections)."""
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
for name in ("kernel_stats.csv", "kernel_trace_summary.txt", "pmc_summary.txt", "bench_under_rocprof.json",
             "train_kernel_trace_summary.txt", "train_bench_line.json", "flow_timeline.txt"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join("profiles", f"{tag}_{name}"))
text = open(os.path.join(src, "pmc_summary.txt")).read()
vals = {}
for block in re.split(r"\n  (?=\S)", text):
    if block.startswith("chain_kernel<128, 1, true, true>"):
        for m in re.finditer(r"(\w+)\s+mean=\s*([\d.]+)", block):
            vals.setdefault(m.group(1), float(m.group(2)))
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
rows, d, planes, heads, tokpad = 71680, 128, 256, 4, 288
alg = {"read_attn_plus_x": 2 * rows * d * 4, "read_packed_weights_2_layers": 2 * 8 * d * d * 4 // 2,
       "write_x": rows * d * 4, "write_q_k_vt": 3 * planes * heads * tokpad * 32 * 4}
out = {
    "round": tag,
    "kernel": "chain_kernel<128,GELU,MLP=true,QKV=true>, B=128 frames (71,680 token rows)",
    "source": f"profiles/{tag}_pmc_summary.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over 50 launches",
    "FETCH_SIZE_KB_mean": fetch_kb, "WRITE_SIZE_KB_mean": write_kb,
    "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced (16 B/lane) reads -> x2; "
                  "WRITE_SIZE is exact for 16 B/lane stores (MI355X_MICROARCH.md, HBM section)",
    "chain_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
    "algorithmic_bytes_per_launch": alg, "algorithmic_total": sum(alg.values()),
    "mfma_busy_frac": vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (vals.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024),
}
# attention: K / V^T re-read check: algorithmic reads = q + k + vt once, writes = attn once
avals = {}
for block in re.split(r"\n  (?=\S)", text):
    if block.startswith("attn_kernel<"):
        for m in re.finditer(r"(\w+)\s+mean=\s*([\d.]+)", block):
            avals.setdefault(m.group(1), float(m.group(2)))
if "FETCH_SIZE" in avals:
    alg_read = 3 * planes * heads * tokpad * 32 * 4
    out["attention"] = {"kernel": "attn_kernel, B=128 frames (1024 (plane, head) problems, 9 query tiles each)",
                        "FETCH_SIZE_KB_mean": avals["FETCH_SIZE"], "WRITE_SIZE_KB_mean": avals.get("WRITE_SIZE"),
                        "read_bytes_per_launch_corrected": int(2 * avals["FETCH_SIZE"] * 1024),
                        "algorithmic_read_bytes": alg_read,
                        "read_ratio": round(2 * avals["FETCH_SIZE"] * 1024 / alg_read, 3)}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
