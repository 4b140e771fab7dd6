#!/bin/bash
# usage: tools/kstats.sh <tag> <python args...>   -> prints top kernels by total time (run on the GPU box)
tag=${1:?tag}; shift
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
script=$1; shift
case "$script" in /*) ;; *) script=$GRAFT_REPO_ROOT/$script ;; esac
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o k -- python3 "$script" "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/k_kernel_stats.csv")))
print("== $tag")
for r in rows[:10]: print("  %-58s %5s %10.1f us" % (r["Name"][:58], r["Calls"], float(r["AverageNs"])/1e3))
PY
