mkdir -p gpurun_out/r5; L=gpurun_out/r5/t15.log; : > $L
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "other_model_dims or random_configurations" 2>&1 | tail -3 >> $L
python - >> $L 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import bench, torch
for tag, over in (("d64_h2", dict(model_dim=64, num_head=2)), ("d32_h1", dict(model_dim=32, num_head=1)), ("d96_h3", dict(model_dim=96, num_head=3))):
    r = bench.batch_sweep(dict(bench.C3, **over, batch=128), torch.device("cuda:0"), (128,), steps=20, warmup=5)
    print(tag, r["value"], r["dominant_frac"], r["dominant"])
PY
cat $L
