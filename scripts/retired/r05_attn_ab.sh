#!/bin/bash
# Round 5, prefill attention A/B on ONE box: score-chain variants 1 (reference rounding chain, 16-row kernel), 3 (f32 chain, 16-row kernel),
# 4 (f32 chain, 32-row kernel, AHA_ATTN32_WAVES = 4 / 8) -- stand-alone text kernel (scripts/bench_attn.py) and the cfg 3 bench's kernels
# under rocprofv3 (ViT 96/80 and text instantiations).  Output: gpurun_out/r05_attn_ab.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/r05_attn_ab.txt
: > $out
for v in "1 0" "3 0" "4 4" "4 8" "1 0" "4 4"; do
  set -- $v
  echo "== smx $1 waves32 $2" >> $out
  AHA_ATTN_SMX=$1 AHA_ATTN32_WAVES=$2 timeout 300 python scripts/bench_attn.py 2>&1 | grep ms/launch >> $out
  AHA_ATTN_SMX=$1 AHA_ATTN32_WAVES=$2 timeout 300 python scripts/bench_attn.py 1542 40980 2>&1 | grep ms/launch >> $out
done
for v in "1 0" "4 4" "3 0" "4 8"; do
  set -- $v
  echo "== rocprof bench smx $1 waves32 $2" >> $out
  rm -rf gpurun_out/prof_ab
  AHA_ATTN_SMX=$1 AHA_ATTN32_WAVES=$2 bash scripts/prof_kernels.sh ab bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "attn_prefill" | cut -c1-140 >> $out
  grep -h "prefill_ms" gpurun_out/prof_ab/run.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench line: value', d['value'], 'prefill_ms', d.get('prefill_ms'))" >> $out 2>&1
done
cat $out
