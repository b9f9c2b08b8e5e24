#!/bin/bash
# SQ / GRBM counters of the prefill attention kernel forms at S = 8192 FULL attention (32 heads x head_dim 128), one counter group per pass, no
# tracing domains next to --pmc.  usage (GPU box): bash scripts/r06_attn_pmc.sh <form: 16 | 64 | 65>   -> gpurun_out/r06_attn_pmc_<form>.txt
# GRBM_GUI_ACTIVE / kernel time = the effective clock under the kernel (MI355X_MICROARCH.md "DVFS give-back"); the kernel time is taken from
# an un-profiled launch of the same process layout (AHA_ATTN_TIME), printed last.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
form=$1
export AHA_ATTN_TIME=3 AHA_ATTN_FORM=$form FORMS=$form
rm -rf $R/gpurun_out/pmc_attn6_${form}_*
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_attn6_${form}_$i -o pmc -- python $R/scripts/attn64_ab.py 8192:0 > $R/gpurun_out/pmc_attn6_${form}_$i.log 2>&1 || true
done
python - $form <<'PY' > $R/gpurun_out/r06_attn_pmc_$form.txt
import csv, glob, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = {}
for f in glob.glob(os.path.join(R, "gpurun_out", f"pmc_attn6_{sys.argv[1]}_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_prefill" not in r["Kernel_Name"]:
            continue
        d = acc.setdefault(r["Counter_Name"], {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
print("form", sys.argv[1], "S = 8192 full, 32 q heads / 8 kv heads, head_dim 128")
for k, d in sorted(acc.items()):
    v = list(d.values())
    print(f"{k}: {len(v)} dispatches, mean per dispatch {sum(v)/len(v):.5g}")
PY
timeout 100 python $R/scripts/attn64_ab.py 8192:0 2>&1 | grep "ms/launch" | tail -2 >> $R/gpurun_out/r06_attn_pmc_$form.txt
cat $R/gpurun_out/r06_attn_pmc_$form.txt
