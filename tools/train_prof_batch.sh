# usage: tools/train_prof_batch.sh NAME BATCH -- kernel trace of the training step at another batch (tile-quantisation experiments)
set -u
NAME=$1; B=$2
REPO=$(pwd); OUT=$REPO/gpurun_out/$NAME; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -- python3 "$REPO/tools/train_bench.py" --only hip --steps 10 --warmup 3 --batch $B > "$OUT/train_trace.log" 2>&1
cd "$REPO"
python3 tools/train_step_breakdown.py "$OUT/train_trace" > "$OUT/train_kernel_trace_summary.txt" 2>&1
find "$OUT" -name "*.csv" -size +1M -delete
head -14 "$OUT/train_kernel_trace_summary.txt"
