set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r06_train64; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -- python3 "$REPO/tools/train_bench.py" --only hip --steps 10 --warmup 3 --batch 64 > "$OUT/train_trace.log" 2>&1
cd "$REPO"
python3 tools/train_step_breakdown.py "$OUT/train_trace" > "$OUT/train_kernel_trace_summary.txt" 2>&1
find "$OUT" -name "*.csv" -size +1M -delete
head -30 "$OUT/train_kernel_trace_summary.txt"
