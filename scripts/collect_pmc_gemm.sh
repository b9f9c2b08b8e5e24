#!/bin/bash
# SQ counters of the four-wave 256^2 GEMM (scripts/bench_gemm.py, shapes "gateup" and "big"), one counter group per pass, no tracing
# domains next to --pmc.  Run on the GPU box:  gpurun -- 'bash scripts/collect_pmc_gemm.sh'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export AHA_GEMM_ONLY=gateup,big
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_gemm_$i -o pmc -- python $R/scripts/bench_gemm.py > $R/gpurun_out/pmc_gemm_$i.log 2>&1 || true
done
python - <<'PY'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = {}
for f in glob.glob(os.path.join(R, "gpurun_out", "pmc_gemm_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256q_kernel" not in r["Kernel_Name"]:
            continue
        key = (r["Counter_Name"], "gateup" if "gemm256q_kernel<4" in r["Kernel_Name"] else "big")
        d = acc.setdefault(key, {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
for k, d in sorted(acc.items()):
    v = list(d.values())
    print(f"{k[1]:7s} {k[0]}: {len(v)} dispatches, mean per dispatch {sum(v)/len(v):.5g}")
PY
