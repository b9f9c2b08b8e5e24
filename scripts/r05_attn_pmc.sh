#!/bin/bash
# SQ / TCC counters of the prefill attention kernel (scripts/bench_attn.py at S = 8192 causal), one counter group per pass, no tracing
# domains next to --pmc.  usage (GPU box): bash scripts/r05_attn_pmc.sh <smx: 0 | 1 | 3> [tag]   -> gpurun_out/r05_attn_pmc_<smx>_<tag>.txt
# (the second argument was the block shape of the round's retired 32-row experiment; it only names the output now)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
smx=$1; w=${2:-0}
export AHA_ATTN_TIME=1 AHA_ATTN_SMX=$smx AHA_ATTN32_WAVES=$w
tag=${smx}_${w}
rm -rf $R/gpurun_out/pmc_attn_${tag}_*
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_attn_${tag}_$i -o pmc -- python $R/scripts/bench_attn.py 8192 > $R/gpurun_out/pmc_attn_${tag}_$i.log 2>&1 || true
done
python - $tag <<'PY' > $R/gpurun_out/r05_attn_pmc_$tag.txt
import csv, glob, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = {}
for f in glob.glob(os.path.join(R, "gpurun_out", f"pmc_attn_{sys.argv[1]}_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_prefill" not in r["Kernel_Name"]:
            continue
        d = acc.setdefault(r["Counter_Name"], {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
print("variant", sys.argv[1])
for k, d in sorted(acc.items()):
    v = list(d.values())
    print(f"{k}: {len(v)} dispatches, mean per dispatch {sum(v)/len(v):.5g}")
PY
cat $R/gpurun_out/r05_attn_pmc_$tag.txt; grep -h "ms/launch" $R/gpurun_out/pmc_attn_${tag}_1.log | tail -1
