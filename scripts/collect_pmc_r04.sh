#!/bin/bash
# Round 4: SQ counters of the prefill attention kernel with the score chain's scale multiply in the vector ALU (AHA_ATTN_SMX=0) and on the
# matrix pipe (1), text geometry at S = 8192 (scripts/bench_attn.py); one counter group per pass, no tracing domains next to --pmc.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export AHA_ATTN_TIME=1
for smx in 0 1; do
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    AHA_ATTN_SMX=$smx timeout 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_attn_smx${smx}_$i -o pmc -- python $R/scripts/bench_attn.py 8192 > $R/gpurun_out/pmc_attn_smx${smx}_$i.log 2>&1 || true
  done
done
python - <<'PY'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for smx in (0, 1):
    acc = {}
    for f in glob.glob(os.path.join(R, "gpurun_out", f"pmc_attn_smx{smx}_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_prefill_kernel" not in r["Kernel_Name"]:
                continue
            d = acc.setdefault(r["Counter_Name"], {})
            d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    print(f"AHA_ATTN_SMX={smx}")
    for k, d in sorted(acc.items()):
        v = list(d.values())
        print(f"  {k}: {len(v)} dispatches, mean per dispatch {sum(v)/len(v):.5g}")
PY
