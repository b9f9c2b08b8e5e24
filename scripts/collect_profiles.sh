#!/bin/bash
# gpurun_out/ (merged back from the GPU box by gpurun) -> the tracked summaries of the round (ROUND=r05 by default) under profiles/.
# Run in the build container.
set -e
cd "$(dirname "$0")/.."
R=${ROUND:-r06}
python scripts/stats_to_md.py gpurun_out/prof_cfg3 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 32 --warmup 4 --no-cpu-baseline ($R)" 32 > profiles/${R}_cfg3_kernel_stats.md
for w in qwen3-0.6b:cfg2 qwen3-asr:cfg4 qwen3vl8b-cfg5:cfg5; do
  [ -d gpurun_out/prof_${w%%:*} ] && python scripts/stats_to_md.py gpurun_out/prof_${w%%:*} "rocprofv3 --kernel-trace --stats -- python bench.py --workload ${w%%:*} --steps 16 --warmup 2 --no-cpu-baseline ($R, BASELINE ${w##*:})" 30 > profiles/${R}_${w##*:}_kernel_stats.md
done
ROUND=$R python scripts/pmc_summary.py > /dev/null
for w in default qwen3vl8b-cfg5 qwen3-0.6b qwen3-asr qwen3vl8b-text qwen3vl8b-cfg5-tp qwen3vl8b-cfg5-cp qwen3vl8b-video; do
  [ -s gpurun_out/bench_$w.json ] && cp gpurun_out/bench_$w.json profiles/${R}_bench_$w.json
done
for f in gemm_ours gemm_vendor attn_prefill attn64_ab attn_decode gemv; do
  [ -s gpurun_out/$f.txt ] && grep -v amdgpu.ids gpurun_out/$f.txt > profiles/${R}_$f.txt
done
ls -la profiles/${R}_*
