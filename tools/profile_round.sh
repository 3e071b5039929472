#!/bin/bash
# Round profile (run on the GPU box through gpurun):  bash tools/profile_round.sh r01
#  1. rocprofv3 --kernel-trace --stats of the SAME command the bench line comes from
#  2. separate --pmc passes (never combined with tracing): FETCH_SIZE / WRITE_SIZE / SQ+GRBM for the
#     dominant kernel, via tools/prof_kernels.py (50 launches of the full chain kernel)
# Summaries land in gpurun_out/<tag>/ ; copy the ones to keep into profiles/.
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 "$REPO/bench.py" --steps 20 --warmup 5 --headline-only > "$OUT/bench_trace.log" 2>&1
export AFT_ONLY=chain AFT_REPS=50 AFT_FWD=1
i=0
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
  "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES TCP_TOTAL_CACHE_ACCESSES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -- python3 "$REPO/tools/prof_kernels.py" > "$OUT/pmc$i.log" 2>&1
done
unset AFT_ONLY
export AFT_ONLY=attention,upsample,tail
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d "$OUT/pmc_attn" -- python3 "$REPO/tools/prof_kernels.py" > "$OUT/pmc_attn.log" 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_attn_fetch" -- python3 "$REPO/tools/prof_kernels.py" > "$OUT/pmc_attn_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_attn_write" -- python3 "$REPO/tools/prof_kernels.py" > "$OUT/pmc_attn_write.log" 2>&1
# training step: kernel trace of the HIP-encoder path + the A/B line against PyTorch-ROCm autograd
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -- python3 "$REPO/tools/train_bench.py" --only hip --steps 10 --warmup 3 > "$OUT/train_trace.log" 2>&1
python3 "$REPO/tools/train_bench.py" --steps 20 --warmup 5 2>/dev/null | grep "^{" | tail -1 > "$OUT/train_bench_line.json"
cd "$REPO"
python3 tools/train_step_breakdown.py "$OUT/train_trace" > "$OUT/train_kernel_trace_summary.txt" 2>&1
python3 tools/summarize_prof.py "$OUT/bench_trace" > "$OUT/kernel_trace_summary.txt" 2>&1
python3 tools/summarize_prof.py "$OUT"/pmc[0-9] "$OUT/pmc_attn" "$OUT/pmc_attn_fetch" "$OUT/pmc_attn_write" > "$OUT/pmc_summary.txt" 2>&1
cp "$OUT"/bench_trace/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
find "$OUT" -name "*.csv" -size +1M -delete
grep -h "^{\"metric\"" "$OUT/bench_trace.log" | tail -1 > "$OUT/bench_under_rocprof.json"
cat "$OUT/kernel_trace_summary.txt"; grep -A12 "chain_kernel" "$OUT/pmc_summary.txt" | head -80
