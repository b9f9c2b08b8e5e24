#!/bin/bash
# Round 5 (round-4 verdict, next-round item 2): fabric traffic and L2 hit rate of the gate+up / qkv prefill GEMMs, per tiling variant, on ONE
# box.  Variants: a = automatic plan (192-column tiles, fifth fragment row), b = 192-column tiles without the fifth row, c = 256^2 tiles;
# AHA_GEMM_GROUP = band width of the tile order.  One counter group per pass, no tracing domains next to --pmc; timing from un-profiled runs.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export AHA_GEMM_ONLY=${SHAPES:-gateup,qkv}
out=$R/gpurun_out/r05_pmc_gemm.txt
: > $out
run() {  # tag, env...
  tag=$1; shift
  echo "== $tag: $*" >> $out
  env "$@" timeout 120 python $R/scripts/bench_gemm.py 2>&1 | grep -E "TFLOP" >> $out
  env "$@" timeout 120 python $R/scripts/bench_gemm.py 2>&1 | grep -E "TFLOP" >> $out
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/pmc_g_${tag}_$i
    env "$@" timeout 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_g_${tag}_$i -o pmc -- python $R/scripts/bench_gemm.py > $R/gpurun_out/pmc_g_${tag}_$i.log 2>&1 || true
  done
  python - $tag <<'PY' >> $out
import csv, glob, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = {}
for f in glob.glob(os.path.join(R, "gpurun_out", f"pmc_g_{sys.argv[1]}_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0][-70:]
        d = acc.setdefault((name, r["Counter_Name"]), {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
for k, d in sorted(acc.items()):
    v = list(d.values())
    print(f"  {k[0]} {k[1]}: {len(v)} dispatches, mean {sum(v)/len(v):.5g}")
PY
}
run a AHA_GEMM_GROUP=8
run b AHA_GEMM_ROW5=0
run c AHA_GEMM_N192=0
run g4 AHA_GEMM_GROUP=4
run g16 AHA_GEMM_GROUP=16
run g0 AHA_GEMM_GROUP=0
cat $out
