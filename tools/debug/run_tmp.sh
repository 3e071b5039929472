mkdir -p gpurun_out/r5; L=gpurun_out/r5/t7.log; : > $L
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "golden or intermediates or ragged or full_size or stale or soak" 2>&1 | tail -3 >> $L
for i in 1 2; do python tools/time_kernels.py upsample tail prologue >> $L 2>&1; AFT_PROLOGUE_NO_UP=1 python tools/time_kernels.py prologue >> $L 2>&1; AFT_CONV_MFMA32=1 python tools/time_kernels.py upsample tail >> $L 2>&1; done
export AFT_LIB_PATH=$PWD/adafortitran_amd/csrc/libaft_hip_diag.so; AFT_STAMPS=1 python tools/time_kernels.py upsample tail 2>&1 | grep "conv stream" | tail -4 >> $L; grep -v amdgpu.ids $L
