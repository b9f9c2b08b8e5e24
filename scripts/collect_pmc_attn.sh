#!/bin/bash
# SQ counters of the prefill attention kernel (scripts/bench_attn.py at S = 8192), one counter group per pass, no tracing
# domains next to --pmc.  Run on the GPU box:  gpurun -- 'bash scripts/collect_pmc_attn.sh'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export AHA_ATTN_TIME=1
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_attn_$i -o pmc -- python $R/scripts/bench_attn.py 8192 > $R/gpurun_out/pmc_attn_$i.log 2>&1 || true
done
python - <<'PY'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = {}
for f in glob.glob(os.path.join(R, "gpurun_out", "pmc_attn_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_prefill_kernel" not in r["Kernel_Name"]:
            continue
        d = acc.setdefault(r["Counter_Name"], {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
for k, d in sorted(acc.items()):
    v = list(d.values())
    print(f"{k}: {len(v)} dispatches, mean per dispatch {sum(v)/len(v):.4g}")
PY
