"""kernel_stats.csv of a `rocprofv3 --kernel-trace --stats` run -> markdown table (profiles/*.md).  usage: stats_to_md.py <dir> <title> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
rows = list(csv.DictReader(open(f[0])))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# " + sys.argv[2] + "\n")
print("| kernel | calls | total_us | avg_us | min_us | max_us | % |\n|---|---|---|---|---|---|---|")
for r in rows[:n]:
    print(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {100*float(r['TotalDurationNs'])/tot:.2f} |")
