#!/usr/bin/env bash
# tools/scale_curve.sh [C3|C2|C5 ...] -- the weak-scaling curve of SURVEY.md 8(e) on ONE node: bench.py --gpus {1,2,4,8}
# back to back for each config (C3 at 8 GPUs = BASELINE config 4, batch 1024 = 128 frames per GPU; C5 at 8 GPUs = BASELINE
# config 5, batch 512 = 64 per GPU), then the table  N | frames/s | ms per step | efficiency = value(N) / (N x value(1)) |
# per-rank device ms min / max | slowest rank | metric all-gather ms.  The one command an 8-GPU driver needs:
#
#     bash tools/scale_curve.sh C3 C5
#
# Each bench.py call self-launches torchrun on 127.0.0.1 (one process per GPU, backend nccl = RCCL) and prints one JSON line;
# this script only collects the lines (gpurun_out/scale_<config>_<N>.json) and does the arithmetic.
# Environment: AFT_SCALE_GPUS="1 2 4 8", AFT_SCALE_STEPS=200, AFT_SCALE_WARMUP=20; AFT_SCALE_STUB=1 = the CPU control-flow stub
# (tests/test_bench_cli.py: no GPU, measures nothing).
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY="${HSA_ENABLE_IPC_MODE_LEGACY:-0}"
GPUS="${AFT_SCALE_GPUS:-1 2 4 8}"
STEPS="${AFT_SCALE_STEPS:-200}"
WARMUP="${AFT_SCALE_WARMUP:-20}"
EXTRA="--headline-only"
[ "${AFT_SCALE_STUB:-0}" = "1" ] && EXTRA="--stub"
OUT="${AFT_SCALE_OUT:-gpurun_out}"
mkdir -p "$OUT"
[ $# -eq 0 ] && set -- C3
for cfg in "$@"; do
    files=()
    for n in $GPUS; do
        f="$OUT/scale_${cfg}_${n}.json"
        python bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARMUP" --config "$cfg" $EXTRA | grep '^{' | tail -1 > "$f"
        [ -s "$f" ] || { echo "scale_curve: bench.py --gpus $n --config $cfg printed no line" >&2; exit 3; }
        files+=("$f")
    done
    python - "$cfg" "${files[@]}" <<'EOF'
import json, sys
cfg, files = sys.argv[1], sys.argv[2:]
lines = [json.load(open(f)) for f in files]
base = next((l for l in lines if l["n_gpus"] == 1), lines[0])
per_gpu = base["value"] / base["n_gpus"]
print(f"# weak scaling, {cfg}: {lines[0]['config']['frames_per_gpu']} frames per GPU per step ({lines[0]['data']})")
print("# N  frames/s  ms/step  efficiency  dev_ms_min  dev_ms_max  slowest  allgather_ms")
for l in lines:
    pr = l.get("per_rank") or {}
    print(f"{l['n_gpus']:<3d} {l['value']:<12.1f} {l['ms_per_step']:<8.4f} {l['value'] / (l['n_gpus'] * per_gpu):<10.4f} "
          f"{pr.get('device_ms_per_step_min', l.get('device_ms_per_step', 0)):<10} {pr.get('device_ms_per_step_max', l.get('device_ms_per_step', 0)):<10} "
          f"{pr.get('slowest_rank', 0):<7} {pr.get('metric_allgather_ms', 0)}")
EOF
done
