"""Summarise the PMC passes of scripts/collect_pmc.sh into profiles/<round>_pmc_traffic_gemv.json (read by bench.py's
roofline.traffic).  FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports HALF the bytes of wide coalesced reads, so
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section).  Per launch = sum over the kernel class / its dispatch count."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r06")


def load(counter):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no counter_collection.csv under gpurun_out/pmc_{counter}")
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            d = per.setdefault(k, {})
            key = r["Dispatch_Id"]
            d[key] = d.get(key, 0.0) + float(r["Counter_Value"])   # summed over XCDs / instances
    return per


def digest(files):   # = bench.py source_digest
    import hashlib
    h = hashlib.sha256()
    for fn in files:
        h.update(fn.encode()); h.update(open(os.path.join(ROOT, "aha_amd", "csrc", fn), "rb").read())
    return h.hexdigest()[:16]


def main():
    fetch, write = load("FETCH_SIZE"), load("WRITE_SIZE")
    cls = [k for k in fetch if "gemv_kernel" in k]
    n = sum(len(fetch[k]) for k in cls)
    fb = sum(sum(fetch[k].values()) for k in cls) * 1024 * 2
    wb = sum(sum(write.get(k, {}).values()) for k in cls) * 1024
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/collect_pmc.sh) -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline, " + ROUND,
        "workload": "qwen3vl8b", "kernel_class": "gemv", "gemv_source_digest": digest(("gemv_body.h", "kernels_gemv.hip", "common.h")),
        "correction": "FETCH_SIZE (KB) x 1024 x 2: gfx950 reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE x 1024 uncalibrated",
        "launches": n, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n,
        "traffic_bytes_per_launch": (fb + wb) / n,
        "per_kernel_fetch_MB_x2": {k[-32:]: round(sum(fetch[k].values()) * 1024 * 2 / len(fetch[k]) / 1e6, 2) for k in cls},
    }
    other = {k[:80]: round(sum(v.values()) * 1024 * 2 / len(v) / 1e6, 3) for k, v in fetch.items()
             if ("attn_decode" in k or "gemm256" in k or "attn_prefill" in k) and len(v)}
    out["other_kernels_fetch_MB_x2_per_launch"] = other
    json.dump(out, open(os.path.join(ROOT, "profiles", ROUND + "_pmc_traffic_gemv.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
