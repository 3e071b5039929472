mkdir -p gpurun_out/r5; L=gpurun_out/r5/t11.log; : > $L
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "column_ranges or golden or full_size or chunks" 2>&1 | tail -5 >> $L
for b in 129 160; do for ns in 1 0; do if [ $ns = 1 ]; then export AFT_CONV_NSPLIT=1; else unset AFT_CONV_NSPLIT; fi; echo "B=$b forced_nsplit1=$ns" >> $L; AFT_BATCH=$b python tools/time_kernels.py upsample tail 2>&1 | grep -v amdgpu >> $L; done; done
cat $L
