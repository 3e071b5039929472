#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/ and derive
profiles/pmc_traffic.json: HBM bytes per launch (gfx950 corrections) and the SQ counters of the dominant kernel (chain),
of attention, and -- as SEPARATE rows -- of the conv head and the conv tail.

The --pmc passes run tools/prof_kernels.py: ONE whole forward (it fills the workspace the replays need) and then AFT_REPS
launches of one kernel class.  Rows are therefore taken per dispatch: the conv head / tail share a kernel symbol, so the
first two conv_stack dispatches of a pass (the forward's head and tail) are skipped and the rest belong to the class the
pass replays."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
for name in ("kernel_stats.csv", "kernel_trace_summary.txt", "plane_kernel_trace_summary.txt", "split_kernel_trace_summary.txt",
             "pmc_summary.txt", "bench_under_rocprof.json", "train_kernel_trace_summary.txt", "train_bench_line.json",
             "c5_kernel_trace_summary.txt", "general_kernel_trace_summary.txt"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join("profiles", f"{tag}_{name}"))


def counters(dirname, kernel_substr, skip_first=0):
    """mean counter values over the dispatches of kernels whose name contains `kernel_substr`, skipping the first
    `skip_first` such dispatches (those of the pass's warm-up forward)"""
    out = defaultdict(list)
    for path in glob.glob(os.path.join(src, dirname, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(path)) if kernel_substr in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        keep = set(ids[skip_first:])
        for r in rows:
            if int(r["Dispatch_Id"]) in keep:
                out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}, max((len(v) for v in out.values()), default=0)


def block(prefix, kernel_substr, skip_first, alg_bytes, note):
    vals, n = {}, 0
    for part in ("fetch", "write", "sq"):
        v, k = counters(f"pmc_{prefix}_{part}", kernel_substr, skip_first)
        vals.update(v)
        n = max(n, k)
    if "FETCH_SIZE" not in vals:
        return None
    rec = {"kernel": note, "launches_averaged": n, "FETCH_SIZE_KB_mean": round(vals["FETCH_SIZE"], 1),
           "WRITE_SIZE_KB_mean": round(vals.get("WRITE_SIZE", 0.0), 1),
           "bytes_per_launch": int((2 * vals["FETCH_SIZE"] + vals.get("WRITE_SIZE", 0.0)) * 1024),
           "algorithmic_bytes_per_launch": alg_bytes}
    rec["ratio_to_algorithmic"] = round(rec["bytes_per_launch"] / alg_bytes, 3)
    if "SQ_INSTS_MFMA" in vals:
        rec["valu_per_mfma"] = round(vals["SQ_INSTS_VALU"] / max(vals["SQ_INSTS_MFMA"], 1), 2)
        if "GRBM_GUI_ACTIVE" in vals:   # GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs
            rec["mfma_busy_frac"] = round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    return rec


rows, d, planes, heads, tokpad, frames = 71680, 128, 256, 4, 288, 128
alg_chain = 2 * rows * d * 4 + 2 * 8 * d * d * 4 // 2 + rows * d * 4 + 3 * planes * heads * tokpad * 32 * 4
out = {
    "round": tag,
    "source": f"profiles/{tag}_pmc_summary.txt + the per-dispatch CSVs of gpurun_out/{tag}/pmc_*: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ "
              "in separate passes (tools/profile_round.sh)",
    "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced (16 B/lane) reads -> x2; "
                  "WRITE_SIZE is exact for 16 B/lane stores (MI355X_MICROARCH.md, HBM section)",
}
chain = block("chain", "chain_kernel<128, 1, true, true>", 5, alg_chain,
              "chain_kernel<128,GELU,MLP=true,QKV=true>, B=128 frames (71,680 token rows)")
if chain:
    out["chain"] = chain
    out["chain_bytes_per_launch"] = chain["bytes_per_launch"]          # bench.py's roofline.traffic
attn = block("attn", "attn_kernel", 6, 3 * planes * heads * tokpad * 32 * 4 + rows * d * 4,
             "attn_kernel, B=128 frames (1024 (plane, head) problems, 9 query tiles each); algorithmic = q + k + v^T read once, attention tiles written once")
if attn:
    out["attention"] = attn
plane = block("plane", "encoder_plane_kernel", 0, 128 * 13644 + 3_950_000,
              "encoder_plane_kernel<128,GELU>: the whole encoder of B=128 as one launch (algorithmic = the forward's compulsory bytes)")
if plane:
    out["encoder_plane"] = plane
# conv stacks: compulsory bytes per frame: head 192 B pilots in + 13,440 B conv_enhanced out (+ 187 KB of weights once per
# launch); tail reads conv_enhanced (13,440) + the linear_2 output (280 tokens x 8 floats x 2 planes = 17,920) and writes 13,440
# round 4: the default grid runs the column-streaming kernels (k_conv_stream.hip) -- conv_stream_kernel<0> = head, <1> = tail, distinct
# symbols -- and the pilot_upsampler product sits in the prologue launch: the head LAUNCH reads the upsampled planes (13,440 B per frame)
# instead of pilots + up_w; the stage's compulsory bytes are unchanged, 13,632 B per frame.
# round 5: conv_stream16_kernel<0> / <1> (16x16x4 matrix phase; 23 KB of operand fragments + tables per launch instead of 19 KB of weights)
head = block("conv_head", "conv_stream16_kernel<0", 1, frames * (13440 + 13440) + 23 * 1024,
             "conv_stream16_kernel<0> HEAD (upsampled planes in, 4 convs, conv_enhanced out), B=128") or \
       block("conv_head", "conv_stream_kernel<0", 1, frames * (13440 + 13440) + 19 * 1024,
             "conv_stream_kernel<0, false> HEAD (upsampled planes in, 4 convs, conv_enhanced out), B=128") or \
       block("conv_head", "conv_stack_kernel", 2, frames * 13632 + 187 * 1024, "conv_stack_kernel<false,true> HEAD (pilot split + Linear 24->1680 + 4 convs), B=128")
tail = block("conv_tail", "conv_stream16_kernel<1", 1, frames * (13440 + 17920 + 13440) + 23 * 1024,
             "conv_stream16_kernel<1> TAIL (fold + residual + 4 convs + complex store), B=128") or \
       block("conv_tail", "conv_stream_kernel<1", 1, frames * (13440 + 17920 + 13440) + 19 * 1024,
             "conv_stream_kernel<1, false> TAIL (fold + residual + 4 convs + complex store), B=128") or \
       block("conv_tail", "conv_stack_kernel", 2, frames * (13440 + 17920 + 13440) + 19 * 1024, "conv_stack_kernel<false,true> TAIL (fold + residual + 4 convs + complex store), B=128")
prol = block("prologue", "prologue_kernel", 0, 3 * 128 * 4 + 128 * 192 + 2 * 3_145_728 + 168_000 + 128 * (13440 + 6720) + 2 * 23 * 1024 + 2 * 19 * 1024,
             "prologue_kernel: channel adapter + weight re-lay (3 MB in, 3 MB out) + pilot_upsampler product over all planes + conv operand fragments, B=128")
if prol:
    out["prologue"] = prol
if head:
    out["conv_head"] = head
if tail:
    out["conv_tail"] = tail
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
