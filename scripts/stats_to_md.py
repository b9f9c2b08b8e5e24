"""kernel_stats.csv of a `rocprofv3 --kernel-trace --stats` run -> markdown table (profiles/*.md).  usage: stats_to_md.py <dir> <title> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
rows = list(csv.DictReader(open(f[0])))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
import hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def digest(files):   # = bench.py source_digest: bench.py attaches these numbers only while the matvec sources are the ones profiled
    h = hashlib.sha256()
    for fn in files:
        h.update(fn.encode()); h.update(open(os.path.join(ROOT, "aha_amd", "csrc", fn), "rb").read())
    return h.hexdigest()[:16]
print("# " + sys.argv[2] + "\n")
print("gemv_source_digest: " + digest(("gemv_body.h", "kernels_gemv.hip", "common.h")) + "\n")
print("| kernel | calls | total_us | avg_us | min_us | max_us | % |\n|---|---|---|---|---|---|---|")
for r in rows[:n]:
    print(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {100*float(r['TotalDurationNs'])/tot:.2f} |")
