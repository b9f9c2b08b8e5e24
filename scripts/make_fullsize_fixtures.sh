#!/bin/bash
# Regenerates the oracle-side fixtures of tests/test_baseline_fullsize_parity_gpu.py (tests/fullsize_cache.py).
# The checkpoints of those tests are drawn by a generator on the GPU, so this runs on the GPU box:
#     gpurun --timeout 1500 -- 'bash scripts/make_fullsize_fixtures.sh'
# It runs the four fixture-backed tests with the LIVE oracle (every step of every free run compared, as through round 5) and leaves
# the fresh fixtures in gpurun_out/fullsize_fixtures/ (merged back by gpurun); in the build container then:
#     cp gpurun_out/fullsize_fixtures/*.npz tests/golden/fullsize/ && python -m pytest tests/test_fullsize_fixtures_cpu.py -q
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
AHA_FULLSIZE_ORACLE=live python -m pytest tests/test_baseline_fullsize_parity_gpu.py -m gpu -x -q --durations=10 \
  -k "cfg3_full_vit or cfg3_decisive or cfg1_cfg2" 2>&1 | tee gpurun_out/make_fullsize_fixtures.log
ls -la gpurun_out/fullsize_fixtures/
