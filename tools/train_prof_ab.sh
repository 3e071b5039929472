# usage: tools/train_prof_ab.sh NAME [ENV=VAL ...]  -- rocprofv3 kernel trace of the training step under the given environment
set -u
NAME=$1; shift
for kv in "$@"; do export "$kv"; done
REPO=$(pwd); OUT=$REPO/gpurun_out/$NAME; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -- python3 "$REPO/tools/train_bench.py" --only hip --steps 10 --warmup 3 > "$OUT/train_trace.log" 2>&1
cd "$REPO"
python3 tools/train_step_breakdown.py "$OUT/train_trace" > "$OUT/train_kernel_trace_summary.txt" 2>&1
find "$OUT" -name "*.csv" -size +1M -delete
head -30 "$OUT/train_kernel_trace_summary.txt"
