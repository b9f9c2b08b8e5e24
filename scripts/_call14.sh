mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -5 gpurun_out/gputest.log
( time timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | grep real; cat gpurun_out/bench_default.json
