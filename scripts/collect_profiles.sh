#!/bin/bash
# gpurun_out/ (merged back from the GPU box by gpurun) -> the tracked round-3 summaries under profiles/.  Run in the build container.
set -e
cd "$(dirname "$0")/.."
python scripts/stats_to_md.py gpurun_out/prof_cfg3 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 32 --warmup 4 --no-cpu-baseline (round 3)" 32 > profiles/r03_cfg3_kernel_stats.md
python scripts/pmc_summary.py > /dev/null
for w in default qwen3vl8b-cfg5 qwen3-0.6b qwen3-asr qwen3vl8b-text qwen3vl8b-cfg5-tp qwen3vl8b-video; do
  [ -s gpurun_out/bench_$w.json ] && cp gpurun_out/bench_$w.json profiles/r03_bench_$w.json
done
for f in gemm_ours gemm_vendor attn_prefill attn_decode gemv; do
  [ -s gpurun_out/$f.txt ] && grep -v amdgpu.ids gpurun_out/$f.txt > profiles/r03_$f.txt
done
ls -la profiles/r03_*
