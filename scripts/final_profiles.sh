#!/bin/bash
# End-of-round measurement set (run on the GPU box through gpurun): kernel stats of the default bench, PMC traffic passes, the
# per-workload bench lines, GEMM / attention microbenchmarks.  Everything lands under gpurun_out/; scripts/collect_profiles.sh (run in
# the build container afterwards) turns it into the tracked summaries under profiles/ (stamped with the matvec source digest that
# bench.py checks before attaching them).
mkdir -p gpurun_out
timeout 300 bash scripts/prof_kernels.sh cfg3 bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/prof_cfg3.txt 2>&1
# round 6: kernel stats of every BASELINE config, not only the headline (round-5 verdict, missing #3): cfg 2, cfg 4, cfg 5 on one GPU
for w in qwen3-0.6b qwen3-asr qwen3vl8b-cfg5; do
  timeout 400 bash scripts/prof_kernels.sh $w bench.py --workload $w --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$w.txt 2>&1
done
timeout 300 bash scripts/collect_pmc.sh > gpurun_out/pmc.log 2>&1
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
for w in qwen3vl8b-cfg5 qwen3-0.6b qwen3-asr qwen3vl8b-text qwen3vl8b-video; do
  timeout 400 python bench.py --workload $w --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
timeout 120 python scripts/bench_gemm.py > gpurun_out/gemm_ours.txt 2>&1
FILLS=weights,zeros timeout 200 python scripts/bench_gemm_vendor.py > gpurun_out/gemm_vendor.txt 2>&1
timeout 200 python scripts/bench_attn.py 2048 8192 40980 > gpurun_out/attn_prefill.txt 2>&1
FORMS=16,65 timeout 300 python scripts/attn64_ab.py 2048:1 4096:1 8192:1 40980:1 2048:0 4096:0 8192:0 > gpurun_out/attn64_ab.txt 2>&1
LENS=1536,8192,40960,131072 timeout 300 python scripts/bench_attn_decode.py > gpurun_out/attn_decode.txt 2>&1
timeout 200 python scripts/bench_gemv.py > gpurun_out/gemv.txt 2>&1
cat gpurun_out/bench_default.json; tail -2 gpurun_out/bench_*.err | head -40
