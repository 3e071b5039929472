mkdir -p gpurun_out/r5; L=gpurun_out/r5/full3.log; : > $L
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $L
python bench.py --verbose-json gpurun_out/r5/bench_verbose3.json > gpurun_out/r5/bench3.json 2> gpurun_out/r5/bench3.err; tail -c 400 gpurun_out/r5/bench3.err >> $L
cat $L
