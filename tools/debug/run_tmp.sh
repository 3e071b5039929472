mkdir -p gpurun_out/r5; L=gpurun_out/r5/t16.log; : > $L
python -m pytest tests/test_train_hip.py -x -q -m gpu -k "one_pass or layer_forward_backward or full_model or dropout" 2>&1 | tail -3 >> $L
for i in 1 2; do
python tools/train_bench.py --only hip --steps 30 --warmup 5 2>/dev/null | grep "^{" | tail -1 | cut -c1-300 >> $L
AFT_LIB_PATH=$PWD/adafortitran_amd/csrc/libaft_hip_fullbar.so python tools/train_bench.py --only hip --steps 30 --warmup 5 2>/dev/null | grep "^{" | tail -1 | cut -c1-300 >> $L
done
cat $L
