mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -4 gpurun_out/gputest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; cat gpurun_out/bench_cfg3.json
timeout 400 python bench.py --workload qwen3vl8b-cfg5 --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err; cat gpurun_out/bench_cfg5.json
timeout 200 python bench.py --workload qwen3-0.6b --no-cpu-baseline > gpurun_out/bench_06b.json 2> gpurun_out/bench_06b.err; cat gpurun_out/bench_06b.json
