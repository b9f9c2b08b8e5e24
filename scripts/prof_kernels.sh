#!/bin/bash
# usage: scripts/prof_kernels.sh <name> <python script + args...>   (run on the GPU box through gpurun)
# rocprofv3 --kernel-trace --stats of the command; prints the top kernels and leaves the csv under gpurun_out/prof_<name>/
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
script=$1; shift; case $script in /*) ;; *) script=$root/$script;; esac
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python $script "$@" > $out/run.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv; log tail:"); print(open(sys.argv[1] + "/run.log").read()[-2000:]); sys.exit(1)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':70s} {'calls':>7s} {'avg us':>9s} {'min us':>9s} {'max us':>9s} {'%':>6s}")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:9.2f} {float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}")
PY
