mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest.log
timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 300 bash scripts/prof_kernels.sh cfg3 bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/prof_cfg3.txt 2>&1
timeout 200 python scripts/bench_gemm.py > gpurun_out/gemm_ours.txt 2>&1
FILLS=weights,zeros,randn timeout 200 python scripts/bench_gemm_vendor.py > gpurun_out/gemm_vendor.txt 2>&1
tail -3 gpurun_out/gputest.log; cat gpurun_out/bench_default.json; cat gpurun_out/gemm_ours.txt gpurun_out/gemm_vendor.txt
