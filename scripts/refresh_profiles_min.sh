#!/bin/bash
# The part of scripts/final_profiles.sh that a kernel change in the prefill path invalidates: kernel stats of the default bench, the default bench
# line (with cpu_baseline), the ViT-heavy workloads.  The PMC matvec passes and the microbenchmarks stay valid while their sources' digest does.
mkdir -p gpurun_out
timeout 300 bash scripts/prof_kernels.sh cfg3 bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/prof_cfg3.txt 2>&1
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
for w in qwen3vl8b-cfg5 qwen3vl8b-video; do
  timeout 400 python bench.py --workload $w --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
cat gpurun_out/bench_default.json | cut -c1-600
