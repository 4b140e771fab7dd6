#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_c4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/bench.py --config 4 --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-events > $O.log 2>&1
python3 $R/tools/timeline.py $(find $O -name 'k_kernel_trace.csv' | head -1)
