#!/bin/bash
# frames/s of the forward path against frames per GPU: bash tools/batch_sweep.sh
for b in 16 32 64 128 256 512; do
  python bench.py --batch $b --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/_bs.json
  python - "$b" <<'PY'
import json, sys
d = json.load(open("/tmp/_bs.json"))
print(f"B={sys.argv[1]:>4}  {d['value']:>9.1f} frames/s  {d['ms_per_step']:.4f} ms/step  chain frac {d['roofline']['frac']}")
PY
done
