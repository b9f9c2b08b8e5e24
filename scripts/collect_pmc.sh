#!/bin/bash
# HBM traffic of the decode kernels from the PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: one counter per
# pass, no tracing domains next to --pmc (gpurun refuses that combination), small step count.  Run on the GPU box:
#   gpurun -- 'bash scripts/collect_pmc.sh'     then     python scripts/pmc_summary.py   (here, on the merged gpurun_out/)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1 || true
  ls $R/gpurun_out/pmc_$c | head -3
done
