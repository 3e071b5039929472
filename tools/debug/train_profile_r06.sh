set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r06; mkdir -p "$OUT"; rm -rf "$OUT/train_trace"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -- python3 "$REPO/tools/train_bench.py" --only hip --steps 10 --warmup 3 > "$OUT/train_trace.log" 2>&1
python3 "$REPO/tools/train_bench.py" --steps 20 --warmup 5 2>/dev/null | grep "^{" | tail -1 > "$OUT/train_bench_line.json"
cd "$REPO"
python3 tools/train_step_breakdown.py "$OUT/train_trace" > "$OUT/train_kernel_trace_summary.txt" 2>&1
find "$OUT/train_trace" -name "*.csv" -size +1M -delete
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
head -4 "$OUT/train_kernel_trace_summary.txt"; cat "$OUT/train_bench_line.json"
