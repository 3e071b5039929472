mkdir -p gpurun_out/r5; L=gpurun_out/r5/t9.log; : > $L
python -m pytest tests/test_train_hip.py tests/test_train_golden.py -x -q -m gpu 2>&1 | tail -15 >> $L
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "module_surface_with" 2>&1 | tail -5 >> $L
cat $L
