#!/bin/bash
# quick A/B of the prefill attention forms on one box: usage r05_attn_quick.sh "<smx>[:waves32[:stg[:pf]]] ..."
cd ${GRAFT_REPO_ROOT:-/root/repo}
variants=$1; shift
for v in $variants; do
  IFS=: read smx w stg pf <<< "$v"
  echo "== smx $smx waves32 ${w:-0} stg ${stg:-0} pf ${pf:-1}"
  export AHA_ATTN_SMX=$smx AHA_ATTN32_WAVES=${w:-0} AHA_ATTN32_STG=${stg:-0} AHA_ATTN32_PF=${pf:-1}
  timeout 300 python scripts/bench_attn.py 2>&1 | grep ms/launch
  timeout 300 python scripts/bench_attn.py 1542 40980 2>&1 | grep ms/launch
done
