mkdir -p gpurun_out/r5; L=gpurun_out/r5/t14.log; : > $L
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "golden or other_model_dims or random_configurations or fewer_than_32 or module_surface_with" 2>&1 | tail -5 >> $L
python - >> $L 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import bench, torch
r = bench.batch_sweep(dict(bench.C3, num_head=8, batch=128), torch.device("cuda:0"), (128,), steps=20, warmup=5)
print("d128_h8_hd16", r)
PY
cat $L
