#!/bin/bash
# SQ / GRBM counters of the 128^2 GEMM kernels inside the BASELINE cfg 4 prefill (Qwen3-ASR: 390 / 406 rows, one block per CU at most), one
# counter group per pass, no tracing domains next to --pmc.  usage (GPU box): bash scripts/r06_gemm_ring_pmc.sh <ring: 1 | 0>
#   -> gpurun_out/r06_gemm_ring_pmc_<ring>.txt   (1 = gemm_glds_ring_kernel, 0 = AHA_GEMM_RING=0: the single-stage gemm_glds_kernel)
# Per kernel name: dispatches and the mean of every counter per dispatch.  GRBM_GUI_ACTIVE = cycles the GPU was busy under the dispatch.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ring=$1
export AHA_GEMM_RING=$ring AHA_GEMM_RING_SPLITK=$ring
rm -rf $R/gpurun_out/pmc_ring_${ring}_*
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_ring_${ring}_$i -o pmc -- python $R/bench.py --workload qwen3-asr --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_ring_${ring}_$i.log 2>&1 || true
done
python - $ring <<'PY' > $R/gpurun_out/r06_gemm_ring_pmc_$ring.txt
import csv, glob, os, re, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = {}
for f in glob.glob(os.path.join(R, "gpurun_out", f"pmc_ring_{sys.argv[1]}_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_glds" not in n:
            continue
        n = re.sub(r"\(.*", "", n.replace("void aha::(anonymous namespace)::", ""))
        d = acc.setdefault(n, {}).setdefault(r["Counter_Name"], {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
print("AHA_GEMM_RING =", sys.argv[1], "-- bench.py --workload qwen3-asr (BASELINE cfg 4): per kernel, mean counter value per dispatch")
for n, cs in sorted(acc.items()):
    nd = max(len(d) for d in cs.values())
    print(f"{n}: {nd} dispatches")
    for k, d in sorted(cs.items()):
        v = list(d.values())
        print(f"    {k}: {sum(v)/len(v):.5g}")
PY
cat $R/gpurun_out/r06_gemm_ring_pmc_$ring.txt | head -80
rm -rf $R/gpurun_out/pmc_ring_${ring}_*   # (the raw per-dispatch csv files are ~40 MB per pass: over gpurun's 64-MiB merge limit)
