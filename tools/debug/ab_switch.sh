# usage: bash tools/debug/ab_switch.sh SWITCH VALUE [train_bench args]  -- the training step with SWITCH=VALUE and without, interleaved
SW=$1; VAL=$2; shift 2
for i in 1 2 3; do
  echo -n "$SW=$VAL: "; env $SW=$VAL python tools/train_bench.py --only hip --steps 20 --warmup 5 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  echo -n "default:   "; python tools/train_bench.py --only hip --steps 20 --warmup 5 "$@" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
