mkdir -p gpurun_out/r5; L=gpurun_out/r5/t13.log; : > $L
python -m pytest tests/test_train_hip.py -x -q -m gpu -k "layer_forward_backward or random_configurations" 2>&1 | tail -12 >> $L
cat $L
