#!/bin/bash
# Round profile (run on the GPU box through gpurun):  bash tools/profile_round.sh r03
#  1. rocprofv3 --kernel-trace --stats of the SAME command the bench line comes from (+ the plane-resident encoder
#     path and the split-precision tier, each as its own trace)
#  2. separate --pmc passes (never combined with tracing): FETCH_SIZE / WRITE_SIZE / SQ+GRBM for the dominant kernel
#     (chain), for attention, and for the conv HEAD and the conv TAIL in separate runs (same kernel symbol, two rows)
#  3. kernel trace of one training step
# Summaries land in gpurun_out/<tag>/ ; tools/update_profiles.py copies the ones to keep into profiles/.
set -u
TAG=${1:-r05}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 "$REPO/bench.py" --steps 20 --warmup 5 --headline-only > "$OUT/bench_trace.log" 2>&1
AFT_ENCODER_PATH=plane rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/plane_trace" -- python3 "$REPO/bench.py" --steps 20 --warmup 5 --headline-only > "$OUT/plane_trace.log" 2>&1
export AFT_REPS=20 AFT_FWD=1
AFT_PRECISION=bf16x3 AFT_FWD=20 AFT_ONLY=none rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/split_trace" -- python3 "$REPO/tools/prof_kernels.py" > "$OUT/split_trace.log" 2>&1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
pmc () {   # pmc <only> <dir tag>
  AFT_ONLY=$1 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_$2_fetch" -- python3 "$REPO/tools/prof_kernels.py" > "$OUT/pmc_$2_fetch.log" 2>&1
  AFT_ONLY=$1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_$2_write" -- python3 "$REPO/tools/prof_kernels.py" > "$OUT/pmc_$2_write.log" 2>&1
  AFT_ONLY=$1 rocprofv3 --pmc $SQ1 --output-format csv -d "$OUT/pmc_$2_sq" -- python3 "$REPO/tools/prof_kernels.py" > "$OUT/pmc_$2_sq.log" 2>&1
}
pmc chain chain
pmc attention attn
pmc upsample conv_head
pmc tail conv_tail
AFT_FWD=6 pmc none prologue     # the forwards' prologue launches (adapter + weight re-lay + upsampler product)
# training step: kernel trace of the HIP path + the A/B line against PyTorch-ROCm autograd
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -- python3 "$REPO/tools/train_bench.py" --only hip --steps 10 --warmup 3 > "$OUT/train_trace.log" 2>&1
python3 "$REPO/tools/train_bench.py" --steps 20 --warmup 5 2>/dev/null | grep "^{" | tail -1 > "$OUT/train_bench_line.json"
# config 5 (240 x 28, 12 layers, d = 256, 64 frames per GPU) and a general-engine shape (d = 512, 8 heads) as their own kernel traces
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5_trace" -- python3 "$REPO/bench.py" --config C5 --steps 5 --warmup 2 --headline-only > "$OUT/c5_trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/general_trace" -- python3 "$REPO/tools/general_forward.py" > "$OUT/general_trace.log" 2>&1
cd "$REPO"
python3 tools/train_step_breakdown.py "$OUT/train_trace" > "$OUT/train_kernel_trace_summary.txt" 2>&1
python3 tools/summarize_prof.py "$OUT/bench_trace" > "$OUT/kernel_trace_summary.txt" 2>&1
python3 tools/summarize_prof.py "$OUT/plane_trace" > "$OUT/plane_kernel_trace_summary.txt" 2>&1
python3 tools/summarize_prof.py "$OUT/c5_trace" > "$OUT/c5_kernel_trace_summary.txt" 2>&1
python3 tools/summarize_prof.py "$OUT/general_trace" > "$OUT/general_kernel_trace_summary.txt" 2>&1
python3 tools/summarize_prof.py "$OUT/split_trace" > "$OUT/split_kernel_trace_summary.txt" 2>&1
python3 tools/summarize_prof.py "$OUT"/pmc_chain_* "$OUT"/pmc_attn_* "$OUT"/pmc_conv_head_* "$OUT"/pmc_conv_tail_* "$OUT"/pmc_prologue_* > "$OUT/pmc_summary.txt" 2>&1
cp "$OUT"/bench_trace/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
find "$OUT" -name "*.csv" -size +1M -delete
grep -h "^{\"metric\"" "$OUT/bench_trace.log" | tail -1 > "$OUT/bench_under_rocprof.json"
cat "$OUT/kernel_trace_summary.txt"; head -12 "$OUT/plane_kernel_trace_summary.txt"; head -14 "$OUT/split_kernel_trace_summary.txt"
