cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/t10
for v in nostore nomfma; do
  export AFT_LIB_PATH=$R/adafortitran_amd/csrc/libaft_hip_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t10/tr_$v -- python3 tools/train_bench.py --only hip --steps 6 --warmup 2 > gpurun_out/t10/log_$v.txt 2>&1
  python3 tools/train_step_breakdown.py gpurun_out/t10/tr_$v --timeline > gpurun_out/t10/summary_$v.txt 2>&1
  rm -rf gpurun_out/t10/tr_$v
done
