mkdir -p gpurun_out/r5; L=gpurun_out/r5/t5.log; : > $L
for i in 1 2 3; do for v in "" _sl8 _sp _sl60; do echo "variant [$v]" >> $L; AFT_LIB_PATH=$PWD/adafortitran_amd/csrc/libaft_hip$v.so python tools/time_kernels.py upsample tail >> $L 2>&1; done; done
grep -v amdgpu.ids $L
