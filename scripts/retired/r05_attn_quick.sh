#!/bin/bash
# quick A/B of the prefill attention forms on one box: usage r05_attn_quick.sh "<smx>[:waves32] ..." [lengths...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
variants=$1; shift
for v in $variants; do
  smx=${v%%:*}; w=0; [[ $v == *:* ]] && w=${v##*:}
  echo "== smx $smx waves32 $w"
  AHA_ATTN_SMX=$smx AHA_ATTN32_WAVES=$w timeout 300 python scripts/bench_attn.py 2>&1 | grep ms/launch
  AHA_ATTN_SMX=$smx AHA_ATTN32_WAVES=$w timeout 300 python scripts/bench_attn.py "${@:-1542 40980}" 2>&1 | grep ms/launch
done
