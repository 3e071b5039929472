# A/B of the plain GEMMs' output-tile height inside the training step and the general engine's forward (switch AFT_GEMM_BM)
for i in 1 2; do
for bm in 64 ""; do
  echo "AFT_GEMM_BM=[$bm]"
  AFT_GEMM_BM=$bm python tools/train_bench.py --only hip --steps 20 --warmup 5 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if 'ms' in k})"
done; done
AFT_GEMM_BM=64 python tools/general_forward.py 2>&1 | tail -1
python tools/general_forward.py 2>&1 | tail -1
