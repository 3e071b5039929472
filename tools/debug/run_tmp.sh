mkdir -p gpurun_out/r5; L=gpurun_out/r5/full2.log; : > $L
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $L
python bench.py --verbose-json gpurun_out/r5/bench_verbose2.json > gpurun_out/r5/bench2.json 2> gpurun_out/r5/bench2.err; tail -c 400 gpurun_out/r5/bench2.err >> $L
cat $L
