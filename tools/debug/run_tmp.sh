mkdir -p gpurun_out/r5; L=gpurun_out/r5/full4.log; : > $L
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $L
python bench.py --verbose-json gpurun_out/r5/bench_verbose4.json > gpurun_out/r5/bench4.json 2> gpurun_out/r5/bench4.err; tail -c 300 gpurun_out/r5/bench4.err >> $L
python __graft_entry__.py smoke 2>&1 | tail -2 >> $L
cat $L
