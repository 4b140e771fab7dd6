#!/bin/bash
# usage (GPU box): tools/prof_timeline.sh <tag> [ENSLAM_LIB path]  -> gpurun_out/tl_<tag>.txt (one graph replay, node by node)
# and gpurun_out/ks_<tag>.csv (rocprofv3 --stats kernel summary) of the default bench command (config 2 only).
tag=$1; lib=$2
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
if [ -n "$lib" ]; then export ENSLAM_LIB=$R/$lib ENSLAM_LIB_ALLOW_MISSING=1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-kernel-events --no-api > $O.log 2>&1
trace=$(find $O -name 'k_kernel_trace.csv' | head -1); stats=$(find $O -name 'k_kernel_stats.csv' | head -1)
python3 $R/tools/timeline.py $trace > $R/gpurun_out/tl_$tag.txt
cp $stats $R/gpurun_out/ks_$tag.csv
cat $R/gpurun_out/tl_$tag.txt
