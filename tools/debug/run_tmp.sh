mkdir -p gpurun_out/r5; L=gpurun_out/r5/t17.log; : > $L
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "tall_planes or row_streaming or config5 or C5" 2>&1 | tail -6 >> $L
for i in 1 2; do AFT_CONFIG=C5 python tools/time_kernels.py upsample tail 2>&1 | grep -v amdgpu >> $L; AFT_CONV_MFMA32=1 AFT_CONFIG=C5 python tools/time_kernels.py upsample tail 2>&1 | grep -v amdgpu >> $L; done
cat $L
